// pinn_mlp_core.h -- device-side building blocks of the fused MLP chain for gfx950.
//
// Layout ("feature-major chain"): one wave owns 32 rows of the time series.  Every
// activation tensor of the net lives in that wave's registers TRANSPOSED, as blocks of
// 32 features x 32 rows in the C/D layout of v_mfma_f32_32x32x2_f32:
//     lane l = (hh = l >> 5, n = l & 31) holds, in register r of block fb,
//     feature fb*32 + (r & 3) + 8*(r >> 2) + 4*hh   of row n.
// With that layout the accumulator of layer l is, register for register, the B operand of
// layer l+1 (k-pair of MFMA step r = features {.., ..+4}), so the whole chain
//     x -> tanh(W0 x) -> ... -> heads
// runs without moving an activation through LDS or HBM.  Only weights stream: each
// H x 32 (forward) or 32 x H (backward) slab of a torch-layout [out, in] matrix is staged
// global -> registers -> LDS once per workgroup (4 waves = 128 rows share it) in a
// two-buffer pipeline, and read back as MFMA A fragments.
//
// Exact fp32: v_mfma_f32_32x32x2_f32 is a k-ordered fmaf chain (no reduced precision), so
// results agree with the reference's float32 CPU path to summation-order noise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pinn_hip.h"

namespace pinn {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;        // 4 waves, one per SIMD (the chain needs > 256 registers/lane)
constexpr int kWaveRows = 32;
constexpr int kTileRows = 128;       // rows per workgroup tile
constexpr int kChunkBytes = 32768;   // one weight slab in LDS
constexpr int kMaxChunks = 160;
constexpr int kMaxDrop = 9;

// ---------------------------------------------------------------------------------------
// flat parameter buffer offsets (floats), state_dict order (include/pinn_hip.h)
// ---------------------------------------------------------------------------------------
struct ParamLayout {
  int H, nh;
  __host__ __device__ long long w0() const { return 0; }
  __host__ __device__ long long b0() const { return 8LL * H; }
  __host__ __device__ long long w(int l) const { return 9LL * H + (long long)(l - 1) * ((long long)H * H + H); }  // l >= 1
  __host__ __device__ long long b(int l) const { return w(l) + (long long)H * H; }
  __host__ __device__ long long wp() const { return 9LL * H + (long long)(nh - 1) * ((long long)H * H + H); }
  __host__ __device__ long long bp() const { return wp() + H; }
  __host__ __device__ long long wv0() const { return bp() + 4; }   // every tensor starts 16-B aligned
  __host__ __device__ long long bv0() const { return wv0() + (long long)(H / 2) * H; }
  __host__ __device__ long long wv1() const { return bv0() + H / 2; }
  __host__ __device__ long long bv1() const { return wv1() + (long long)(H / 4) * (H / 2); }
  __host__ __device__ long long wv2() const { return bv1() + H / 4; }
  __host__ __device__ long long bv2() const { return wv2() + H / 4; }
  __host__ __device__ long long total() const { return bv2() + 4; }
};

// ---------------------------------------------------------------------------------------
// dropout source (device copy of pinn_dropout_t)
// ---------------------------------------------------------------------------------------
struct DropDev {
  int mode;
  unsigned thr[kMaxDrop];     // 16-bit drop thresholds: keep iff draw16 >= thr
  float scale[kMaxDrop];      // 1 / (1 - p) in float32 (torch: noise.div_(1 - p))
  unsigned seed_lo, seed_hi;
  unsigned stream;
  long long row_offset;
  const unsigned* bits;
  int words;                  // uint32 words per row-pass of injected masks
  int nb;                     // words per hidden-layer mask (H / 32)
};

__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = (unsigned)(p1 >> 32) ^ c1 ^ k0;
    const unsigned n2 = (unsigned)(p0 >> 32) ^ c3 ^ k1;
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// 16 keep bits (bit r <-> register r) of feature block `fb` of dropout module `layer` for this
// lane's row.  Branch-free so it can be scheduled into MFMA shadows: eval mode is thr = 0
// (every draw kept).  kBits = true (parity-test kernels only) reads injected bit masks instead.
template <bool kBits>
__device__ __forceinline__ unsigned keep_bits(const DropDev& d, unsigned thr, int layer, int fb, int hh, long long grow,
                                              long long lrow, long long n_rows, unsigned pass) {
  if (kBits) {
    const unsigned word = d.bits[((long long)pass * n_rows + lrow) * d.words + layer * d.nb + fb];
    unsigned keep = 0;
#pragma unroll
    for (int r = 0; r < 16; ++r) keep |= ((word >> (8 * (r >> 2) + 4 * hh + (r & 3))) & 1u) << r;
    return thr == 0 ? 0xFFFFu : keep;
  }
  unsigned keep = 0;
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    unsigned o[4];
    philox4x32_10((unsigned)grow, (unsigned)((unsigned long long)grow >> 32),
                  ((unsigned)layer << 16) | ((unsigned)fb << 2) | ((unsigned)hh << 1) | (unsigned)c, d.stream + pass,
                  d.seed_lo, d.seed_hi, o);
#pragma unroll
    for (int w = 0; w < 4; ++w) {
#pragma unroll
      for (int s = 0; s < 2; ++s) {
        const unsigned draw = (o[w] >> (16 * s)) & 0xFFFFu;
        const int r = 4 * (2 * c + (w >> 1)) + 2 * (w & 1) + s;
        keep |= (draw >= thr ? 1u : 0u) << r;
      }
    }
  }
  return keep;
}

// tanh in float32: odd polynomial below 1/8 (abs err < 2e-10), 1 - 2/(e^{2x}+1) above
// (v_exp_f32 + v_rcp_f32; abs err ~1e-7).
__device__ __forceinline__ float tanh_f32(float x) {
  const float ax = fabsf(x);
  const float x2 = x * x;
  const float poly = x * (1.0f + x2 * (-0.33333334f + x2 * (0.13333334f + x2 * (-0.053968254f))));
  const float e = __builtin_amdgcn_exp2f(ax * 2.8853900817779268f);   // e^{2|x|}
  const float t = 1.0f - 2.0f * __builtin_amdgcn_rcpf(e + 1.0f);
  const float big = copysignf(t, x);
  return ax < 0.125f ? poly : big;
}

// ---------------------------------------------------------------------------------------
// weight-slab pipeline: global -> registers -> LDS, two buffers, one barrier per slab
// ---------------------------------------------------------------------------------------
struct ChunkDesc {
  unsigned off;      // float offset of the slab's first element in the flat parameter buffer
  unsigned short ld; // row stride (floats) of the source matrix
  unsigned char kind;   // 0: forward slab  [rows][32 cols], 16-B chunks XOR-swizzled by (row>>1)&7
                        // 1: backward slab [32 rows][ld cols], contiguous copy
  unsigned char np;     // 16-byte pieces per thread (slab bytes / 4096)
};

struct Pipe {
  const float* params;
  const ChunkDesc* tab;   // in LDS
  char* lds;              // 2 * kChunkBytes
  int n, ci;              // number of slabs in the cycle, slab being computed
  f32x4 regs[8];
  ChunkDesc pending;

  __device__ __forceinline__ void issue(int idx) {
    pending = tab[idx];
    const int tid = threadIdx.x;
    const float* g;
    long long gstride;
    if (pending.kind == 0) {
      g = params + pending.off + (long long)(tid >> 3) * pending.ld + (tid & 7) * 4;
      gstride = 32LL * pending.ld;
    } else {
      g = params + pending.off + tid * 4;
      gstride = 1024;
    }
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < pending.np) regs[p] = *reinterpret_cast<const f32x4*>(g + p * gstride);
  }
  __device__ __forceinline__ void commit(int buf) {
    const int tid = threadIdx.x;
    char* dst = lds + buf * kChunkBytes;
    if (pending.kind == 0)
      dst += (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
    else
      dst += tid * 16;
#pragma unroll
    for (int p = 0; p < 8; ++p)
      if (p < pending.np) *reinterpret_cast<f32x4*>(dst + p * 4096) = regs[p];
  }
  // stage slab 0 synchronously, then start fetching slab 1
  __device__ __forceinline__ void prime() {
    ci = 0;
    issue(0);
    commit(0);
    __syncthreads();
    issue(n > 1 ? 1 : 0);
  }
  __device__ __forceinline__ const char* cur() const { return lds + (ci & 1) * kChunkBytes; }
  // call after the MFMAs of slab ci: publishes slab ci+1 and starts fetching slab ci+2
  __device__ __forceinline__ void advance() {
    commit((ci + 1) & 1);
    __syncthreads();
    ++ci;
    int nxt = ci + 1;
    // the slab index cycles with period n; ci itself keeps counting so buffer parity alternates
    issue(nxt % n);
  }
};

// chunk cycle of one forward pass; returns the count.  tab must hold kMaxChunks entries.
__device__ __forceinline__ int build_forward_chunks(ChunkDesc* tab, const ParamLayout& L, int at) {
  const int H = L.H, NB = H / 32;
  int k = at;
  for (int l = 1; l < L.nh; ++l)
    for (int kb = 0; kb < NB; ++kb) tab[k++] = ChunkDesc{(unsigned)(L.w(l) + kb * 32), (unsigned short)H, 0, (unsigned char)(H / 32)};
  for (int kb = 0; kb < NB; ++kb) tab[k++] = ChunkDesc{(unsigned)(L.wv0() + kb * 32), (unsigned short)H, 0, (unsigned char)(H / 64)};
  for (int kb = 0; kb < NB / 2; ++kb)
    tab[k++] = ChunkDesc{(unsigned)(L.wv1() + kb * 32), (unsigned short)(H / 2), 0, (unsigned char)(H / 128)};
  return k;
}

// chunk cycle of the backward (dgrad) chain, in the order the chain consumes them
__device__ __forceinline__ int build_backward_chunks(ChunkDesc* tab, const ParamLayout& L, int at) {
  const int H = L.H;
  int k = at;
  // d hv1 = Wv1^T d pre_v2 : slabs = 32-row groups of Wv1 [H/4, H/2]
  for (int kb = 0; kb < H / 128; ++kb)
    tab[k++] = ChunkDesc{(unsigned)(L.wv1() + (long long)kb * 32 * (H / 2)), (unsigned short)(H / 2), 1, (unsigned char)(H / 64)};
  // d h_last (variance branch) = Wv0^T d pre_v1 : Wv0 [H/2, H]
  for (int kb = 0; kb < H / 64; ++kb)
    tab[k++] = ChunkDesc{(unsigned)(L.wv0() + (long long)kb * 32 * H), (unsigned short)H, 1, (unsigned char)(H / 32)};
  // hidden layers nh-1 .. 1 : W_l [H, H]
  for (int l = L.nh - 1; l >= 1; --l)
    for (int kb = 0; kb < H / 32; ++kb)
      tab[k++] = ChunkDesc{(unsigned)(L.w(l) + (long long)kb * 32 * H), (unsigned short)H, 1, (unsigned char)(H / 32)};
  return k;
}

// ---------------------------------------------------------------------------------------
// MFMA layer bodies
// ---------------------------------------------------------------------------------------
#define PINN_MFMA(a, b, c) __builtin_amdgcn_mfma_f32_32x32x2f32((a), (b), (c), 0, 0, 0)

template <int NBLK>
__device__ __forceinline__ void zero_blocks(f32x16 (&v)[NBLK]) {
#pragma unroll
  for (int mt = 0; mt < NBLK; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) v[mt][r] = 0.0f;
}

// Per-row context of a lane
struct RowCtx {
  int lane, hh;
  long long grow;     // global row index (Philox counter)
  long long lrow;     // local row index (injected bit masks)
  long long n_rows;
  unsigned pass;
  int mode;           // PINN_DROP_*
};

// Forward layer with LAZY input activation.
//   out^T[f2][n] += sum_f W[f2][f] h^T[f][n], one forward slab per 32-feature input block kb.
// `prep(kb)` must turn block kb of h from "raw accumulator of the previous layer" into the final
// activation (bias + tanh + dropout, stash, head partial sums ...).  It is called ONE SLAB AHEAD
// of its use, in the same scheduling region as the 128 MFMAs of slab kb-1, so its VALU / memory
// work fills the shadow of those MFMAs instead of idling the matrix pipe between layers.
template <int NBIN, int NBOUT, typename F>
__device__ __forceinline__ void layer_forward_lazy(f32x16 (&acc)[NBOUT], const f32x16 (&h)[NBIN], Pipe& pipe, int lane,
                                                   F&& prep) {
  const int hh = lane >> 5, i = lane & 31;
  const int sw = (i >> 1) & 7;
  const int base = i * 128;
  prep(0);
#pragma unroll
  for (int kb = 0; kb < NBIN; ++kb) {
    if (kb + 1 < NBIN) prep(kb + 1);
    const char* buf = pipe.cur() + base;
    // A fragments (16 B = four k-steps of one 32x32 output tile) are fetched kAhead groups before use
    constexpr int kGroups = 4 * NBOUT, kAhead = 2;
    f32x4 afrag[kAhead + 1];
#pragma unroll
    for (int p = 0; p < kAhead; ++p)
      afrag[p] = *reinterpret_cast<const f32x4*>(buf + (((2 * (p / NBOUT) + hh) ^ sw) << 4) + (p % NBOUT) * 4096);
#pragma unroll
    for (int gi = 0; gi < kGroups; ++gi) {
      const int g = gi / NBOUT, mt = gi % NBOUT;
      if (gi + kAhead < kGroups) {
        const int gn = (gi + kAhead) / NBOUT, mn = (gi + kAhead) % NBOUT;
        afrag[(gi + kAhead) % (kAhead + 1)] = *reinterpret_cast<const f32x4*>(buf + (((2 * gn + hh) ^ sw) << 4) + mn * 4096);
      }
      const f32x4 a = afrag[gi % (kAhead + 1)];
      acc[mt] = PINN_MFMA(a[0], h[kb][4 * g + 0], acc[mt]);
      acc[mt] = PINN_MFMA(a[1], h[kb][4 * g + 1], acc[mt]);
      acc[mt] = PINN_MFMA(a[2], h[kb][4 * g + 2], acc[mt]);
      acc[mt] = PINN_MFMA(a[3], h[kb][4 * g + 3], acc[mt]);
    }
    pipe.advance();
  }
}

// Backward (dgrad) layer with lazy operand preparation:
//   din^T[f][n] += sum_f2 W[f2][f] dpre^T[f2][n]; one backward slab (32 rows of W) per dpre block.
template <int NBK, int NBOUT, typename F>
__device__ __forceinline__ void layer_backward_lazy(f32x16 (&acc)[NBOUT], const f32x16 (&dpre)[NBK], Pipe& pipe, int lane,
                                                    int ld, F&& prep) {
  const int hh = lane >> 5, i = lane & 31;
  prep(0);
#pragma unroll
  for (int kb = 0; kb < NBK; ++kb) {
    if (kb + 1 < NBK) prep(kb + 1);
    const float* buf = reinterpret_cast<const float*>(pipe.cur()) + i;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const float* row = buf + (8 * g + 4 * hh) * ld;
#pragma unroll
      for (int mt = 0; mt < NBOUT; ++mt) {
        acc[mt] = PINN_MFMA(row[mt * 32], dpre[kb][4 * g + 0], acc[mt]);
        acc[mt] = PINN_MFMA(row[mt * 32 + ld], dpre[kb][4 * g + 1], acc[mt]);
        acc[mt] = PINN_MFMA(row[mt * 32 + 2 * ld], dpre[kb][4 * g + 2], acc[mt]);
        acc[mt] = PINN_MFMA(row[mt * 32 + 3 * ld], dpre[kb][4 * g + 3], acc[mt]);
      }
    }
    pipe.advance();
  }
}

// input layer: acc = W0 x^T (bias added by the lazy activation), W0 [H, 8] read straight from global
template <int NBOUT>
__device__ __forceinline__ void layer_input(f32x16 (&acc)[NBOUT], const float* __restrict__ W0, const f32x4& xa,
                                            const f32x4& xb, int lane) {
  const int hh = lane >> 5, i = lane & 31;
  const float x0 = hh ? xa[1] : xa[0], x1 = hh ? xa[3] : xa[2], x2 = hh ? xb[1] : xb[0], x3 = hh ? xb[3] : xb[2];
#pragma unroll
  for (int mt = 0; mt < NBOUT; ++mt) {
    const f32x4 wa = *reinterpret_cast<const f32x4*>(W0 + (mt * 32 + i) * 8);
    const f32x4 wb = *reinterpret_cast<const f32x4*>(W0 + (mt * 32 + i) * 8 + 4);
    f32x16 c;
#pragma unroll
    for (int r = 0; r < 16; ++r) c[r] = 0.0f;
    c = PINN_MFMA(hh ? wa[1] : wa[0], x0, c);
    c = PINN_MFMA(hh ? wa[3] : wa[2], x1, c);
    c = PINN_MFMA(hh ? wb[1] : wb[0], x2, c);
    c = PINN_MFMA(hh ? wb[3] : wb[2], x3, c);
    acc[mt] = c;
  }
}

// one 32-feature block: v = dropout(tanh(v + bias)); returns the 16 keep bits.  Branch-free.
template <bool kBits>
__device__ __forceinline__ unsigned activate_block(f32x16& v, const float* __restrict__ bias32, const DropDev& d,
                                                   const RowCtx& c, int layer, int fb, bool has_drop) {
  const bool drop = has_drop && c.mode != PINN_DROP_NONE;
  const unsigned thr = drop ? d.thr[layer] : 0u;
  const float scale = drop ? d.scale[layer] : 1.0f;
  const unsigned keep = keep_bits<kBits>(d, thr, layer, fb, c.hh, c.grow, c.lrow, c.n_rows, c.pass);
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias32 + 8 * q + 4 * c.hh);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int r = 4 * q + j;
      const float a = tanh_f32(v[r] + b[j]);
      v[r] = ((keep >> r) & 1u) ? a * scale : 0.0f;
    }
  }
  return keep;
}

// tanh only (no dropout module after this layer)
__device__ __forceinline__ void activate_block_tanh(f32x16& v, const float* __restrict__ bias32, int hh) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 b = *reinterpret_cast<const f32x4*>(bias32 + 8 * q + 4 * hh);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[4 * q + j] = tanh_f32(v[4 * q + j] + b[j]);
  }
}

// partial <w, h> over one block (this lane's 16 features)
__device__ __forceinline__ float block_dot(const f32x16& h, const float* __restrict__ w32, int hh, float s) {
#pragma unroll
  for (int q = 0; q < 4; ++q) {
    const f32x4 wv = *reinterpret_cast<const f32x4*>(w32 + 8 * q + 4 * hh);
    s = fmaf(wv[0], h[4 * q + 0], s);
    s = fmaf(wv[1], h[4 * q + 1], s);
    s = fmaf(wv[2], h[4 * q + 2], s);
    s = fmaf(wv[3], h[4 * q + 3], s);
  }
  return s;
}

// element (feature f, row n) of 32-row tile `tile32` of an F-feature tensor lives at ((tile32*F + f)*32 + n):
// one block = 16 dword accesses per lane, each wave-instruction two contiguous 128-B segments
__device__ __forceinline__ float* tiled_block_ptr(float* base, long long tile32, int F, int fb, int lane) {
  return base + (tile32 * F + fb * 32 + 4 * (lane >> 5)) * 32 + (lane & 31);
}
__device__ __forceinline__ void store_block(float* __restrict__ p, const f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) p[((r & 3) + 8 * (r >> 2)) * 32] = v[r];
}
__device__ __forceinline__ void load_block(const float* __restrict__ p, f32x16& v) {
#pragma unroll
  for (int r = 0; r < 16; ++r) v[r] = p[((r & 3) + 8 * (r >> 2)) * 32];
}

// logvar = log(softplus(z) + 1e-6), softplus with torch's threshold 20 (01:432-434)
__device__ __forceinline__ float softplus_f32(float z) { return z > 20.0f ? z : log1pf(expf(z)); }

// ---------------------------------------------------------------------------------------
// One forward pass of the whole net for this wave's 32 rows -> (u, z); both lane halves hold them.
// TRAIN: also parks the keep bits in LDS (keep[(module*NB + block)*64]) and writes the
// post-dropout activations to the tiled stash.
// ---------------------------------------------------------------------------------------
struct StashPtrs {
  float* h;        // [nh][T32][H][32]
  float* v1;       // [T32][H/2][32]
  float* v2;       // [T32][H/4][32]
  long long t32_total;
  long long tile32;
  unsigned short* keep;   // this lane's LDS slot base
};

template <int H, bool TRAIN, bool kBits>
__device__ __forceinline__ void forward_pass(const float* __restrict__ P, const ParamLayout& L, Pipe& pipe, const DropDev& d,
                                             const RowCtx& c, const f32x4& xa, const f32x4& xb, const StashPtrs& st, float& u,
                                             float& z, f32x16 (&v2)[H / 128]) {
  constexpr int NB = H / 32, NB2 = H / 64, NB4 = H / 128;
  const int lane = c.lane, hh = c.hh;
  f32x16 h[NB];
  layer_input<NB>(h, P + L.w0(), xa, xb, lane);
  // hidden layers 1 .. nh-1: while layer l's slabs multiply, layer l-1's raw output is activated block by block
#pragma unroll 1
  for (int l = 1; l < L.nh; ++l) {
    f32x16 acc[NB];
    zero_blocks<NB>(acc);
    const float* bias = P + (l == 1 ? L.b0() : L.b(l - 1));
    layer_forward_lazy<NB, NB>(acc, h, pipe, lane, [&](int kb) {
      const unsigned keep = activate_block<kBits>(h[kb], bias + kb * 32, d, c, l - 1, kb, true);
      if (TRAIN) {
        st.keep[((l - 1) * NB + kb) * 64] = (unsigned short)keep;
        store_block(tiled_block_ptr(st.h + (long long)(l - 1) * st.t32_total * H * 32, st.tile32, H, kb, lane), h[kb]);
      }
    });
#pragma unroll
    for (int mt = 0; mt < NB; ++mt) h[mt] = acc[mt];
  }
  // h = raw output of the last hidden layer; its activation + the predict head ride on the variance head's first layer
  f32x16 v1[NB2];
  zero_blocks<NB2>(v1);
  float up = 0.0f;
  {
    const int ll = L.nh - 1;
    const float* bias = P + (ll == 0 ? L.b0() : L.b(ll));
    layer_forward_lazy<NB, NB2>(v1, h, pipe, lane, [&](int kb) {
      const unsigned keep = activate_block<kBits>(h[kb], bias + kb * 32, d, c, ll, kb, true);
      up = block_dot(h[kb], P + L.wp() + kb * 32, hh, up);
      if (TRAIN) {
        st.keep[(ll * NB + kb) * 64] = (unsigned short)keep;
        store_block(tiled_block_ptr(st.h + (long long)ll * st.t32_total * H * 32, st.tile32, H, kb, lane), h[kb]);
      }
    });
  }
  u = up + __shfl_xor(up, 32, 64) + P[L.bp()];
  zero_blocks<NB4>(v2);
  layer_forward_lazy<NB2, NB4>(v2, v1, pipe, lane, [&](int kb) {
    const unsigned keep = activate_block<kBits>(v1[kb], P + L.bv0() + kb * 32, d, c, L.nh, kb, true);
    if (TRAIN) {
      st.keep[(L.nh * NB + kb) * 64] = (unsigned short)keep;
      store_block(tiled_block_ptr(st.v1, st.tile32, H / 2, kb, lane), v1[kb]);
    }
  });
  float zp = 0.0f;
#pragma unroll
  for (int mt = 0; mt < NB4; ++mt) {
    activate_block_tanh(v2[mt], P + L.bv1() + mt * 32, hh);
    zp = block_dot(v2[mt], P + L.wv2() + mt * 32, hh, zp);
    if (TRAIN) store_block(tiled_block_ptr(st.v2, st.tile32, H / 4, mt, lane), v2[mt]);
  }
  z = zp + __shfl_xor(zp, 32, 64) + P[L.bv2()];
}

}  // namespace pinn
