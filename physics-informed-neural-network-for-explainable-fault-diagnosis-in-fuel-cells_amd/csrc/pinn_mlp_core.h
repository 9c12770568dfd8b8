// pinn_mlp_core.h -- device-side building blocks of the fused MLP chain for gfx950.
//
// Layout ("feature-major chain"): one wave owns 16 rows of the time series.  Every
// activation tensor of the net lives in that wave's registers TRANSPOSED, as blocks of
// 16 features x 16 rows in the C/D layout of v_mfma_f32_16x16x4_f32:
//     lane l = (kq = l >> 4, n = l & 15) holds, in register r of block ib,
//     feature ib*16 + 4*kq + r   of row n.
// With that layout the accumulator of layer l is, register for register, the B operand of
// layer l+1 (MFMA k-step r of input block ib contracts features {r, 4+r, 8+r, 12+r} of the
// block), so the whole chain  x -> tanh(W0 x) -> ... -> heads  runs without moving an
// activation through LDS or HBM.  Only weights stream: each [out x 32] (forward) or [32 x in]
// (backward) slab of a torch-layout [out, in] matrix is staged global -> registers -> LDS once
// per workgroup (4 waves = 64 rows share it) in a two-buffer pipeline, and read back as MFMA A
// fragments (forward: one conflict-free ds_read_b128 = four k-steps).
//
// Occupancy is part of the design: 64 + 64 activation/accumulator registers keep a wave under
// 256 registers, so TWO independent workgroups share a CU (2 waves per SIMD).  Their barriers,
// activation (tanh / Philox) phases and weight staging are uncorrelated, so one workgroup's
// VALU and wait time hides under the other's MFMAs -- hipcc will not interleave the two inside
// a single wave (checked in the ISA), the hardware does it across waves.
//
// Exact fp32: v_mfma_f32_16x16x4_f32 is a k-ordered fmaf chain (no reduced precision), so
// results agree with the reference's float32 CPU path to summation-order noise.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/pinn_hip.h"

namespace pinn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kThreads = 256;        // 4 waves; two such workgroups per CU
constexpr int kWaveRows = 16;
constexpr int kTileRows = 64;        // rows per workgroup tile
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
  __host__ __device__ long long b(int l) const { return l == 0 ? b0() : w(l) + (long long)H * H; }
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

// Small parameter vectors (all biases, the predict and final variance weight rows) are copied into
// LDS once per launch: read per layer between MFMA phases, an L2-latency global load there is an
// exposed stall, an LDS read is ~5x shorter.
struct SmallLayout {
  int H, nh;
  __host__ __device__ int b(int l) const { return l * H; }
  __host__ __device__ int bp() const { return nh * H; }
  __host__ __device__ int bv0() const { return nh * H + 4; }
  __host__ __device__ int bv1() const { return bv0() + H / 2; }
  __host__ __device__ int bv2() const { return bv1() + H / 4; }
  __host__ __device__ int wp() const { return bv2() + 4; }
  __host__ __device__ int wv2() const { return wp() + H; }
  __host__ __device__ int total() const { return wv2() + H / 4; }
};
constexpr int kMaxSmall = 8 * 256 + 4 + 128 + 64 + 4 + 256 + 64;

__device__ __forceinline__ void load_small_params(float* __restrict__ sp, const float* __restrict__ P, const ParamLayout& L) {
  const SmallLayout S{L.H, L.nh};
  const int H = L.H, tid = threadIdx.x;
  for (int l = 0; l < L.nh; ++l)
    for (int i = tid; i < H; i += kThreads) sp[S.b(l) + i] = P[L.b(l) + i];
  for (int i = tid; i < H; i += kThreads) sp[S.wp() + i] = P[L.wp() + i];
  for (int i = tid; i < H / 2; i += kThreads) sp[S.bv0() + i] = P[L.bv0() + i];
  for (int i = tid; i < H / 4; i += kThreads) {
    sp[S.bv1() + i] = P[L.bv1() + i];
    sp[S.wv2() + i] = P[L.wv2() + i];
  }
  if (tid == 0) { sp[S.bp()] = P[L.bp()]; sp[S.bv2()] = P[L.bv2()]; }
  __syncthreads();
}

// pinn_net_range_status's record at the end of d_packed: one word per pack-kernel workgroup (jobs x kRangePackBlocks, rewritten
// by every pack: no atomics, nothing to clear), then one word for the gradient check of the last training call
constexpr int kRangePackBlocks = 64;
constexpr size_t kRangeStatusBytes = 8192;       // >= (18 jobs x 64 + 1) words
unsigned* range_status_words(const pinn_net_t* net);      // pinn_bf16.hip; nullptr where the precision keeps no record

// ---------------------------------------------------------------------------------------
// dropout source (device copy of pinn_dropout_t)
// ---------------------------------------------------------------------------------------
struct DropDev {
  int mode;
  unsigned thr[kMaxDrop];     // 16-bit drop thresholds round(65536 p): keep iff draw16 >= thr (|P(keep) - (1 - p)| <= 2^-17)
  float scale[kMaxDrop];      // 1 / (1 - p) in float32 (torch: noise.div_(1 - p))
  unsigned seed_lo, seed_hi;
  unsigned stream;
  long long row_offset;
  const unsigned* bits;
  int words;                  // uint32 words per row-pass of injected masks
  int nb;                     // words per hidden-layer mask (H / 32)
  unsigned* step_counter;     // training calls only (pinn_dropout_t.d_step_counter): device count of completed optimizer steps = this
                              // call's pass index (Philox stream `stream + pass`, injected masks: pass `pass` of `bits`); or nullptr
};
// pass index of a training call (RowCtx::pass): 0, or the device's step counter for replayed graphs
__device__ __forceinline__ unsigned train_pass(const DropDev& d) {
  return d.step_counter ? __builtin_amdgcn_readfirstlane(*d.step_counter) : 0u;
}

// The round keys are wave-uniform (key + round * Weyl constant: scalar unit) and each counter update is one three-input XOR
// (v_bitop3_b32, truth table 0x96): 20 multiplies + 20 XORs per call.
__device__ __forceinline__ void philox4x32_10(unsigned c0, unsigned c1, unsigned c2, unsigned c3, unsigned k0,
                                              unsigned k1, unsigned (&o)[4]) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * c0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * c2;
    const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), c1, k0 + (unsigned)r * 0x9E3779B9u, 0x96);
    const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), c3, k1 + (unsigned)r * 0xBB67AE85u, 0x96);
    c1 = (unsigned)p1; c3 = (unsigned)p0; c0 = n0; c2 = n2;
  }
  o[0] = c0; o[1] = c1; o[2] = c2; o[3] = c3;
}

// Per-row context of a lane
struct RowCtx {
  int lane, kq;
  long long grow;     // global row index (Philox counter)
  long long lrow;     // local row index (injected bit masks)
  long long n_rows;
  unsigned pass;
  int mode;           // PINN_DROP_*
};

// 8 keep bits of the 32-feature group `fp` (= two 16-feature blocks 2fp, 2fp+1) of dropout
// module `layer` for this lane's row: bit 4*b + r <-> register r of block 2fp + b, i.e. feature
// 32fp + 16b + 4kq + r.  ONE Philox call = eight 16-bit draws = exactly this lane's share of the group:
//     counter = (global_row lo, hi, layer << 16 | fp << 2 | kq, stream + pass), key = seed
//     draw index 4b + r -> word (index >> 1), half (index & 1; 0 = low 16 bits); keep iff draw16 >= thr.
// (Round 2 drew sixteen 8-bit values per call: P(keep) = 205/256 at p = 0.2 under the reference's 1 / (1 - p) scale
// biased every dropout layer's expected output by +0.1 % (+0.26 % at p = 0.4); with 16 bits the bias is < 4e-6.)
// Branch-free: eval mode is thr = 0 (every draw kept).  kBits (parity-test kernels only) reads
// injected bit masks instead.
template <bool kBits>
__device__ __forceinline__ unsigned keep_bits16(const DropDev& d, const RowCtx& c, unsigned thr, int layer, int fp) {
  if (kBits) {
    const unsigned word = d.bits[((long long)c.pass * c.n_rows + c.lrow) * d.words + layer * d.nb + fp];
    const unsigned lo = (word >> (4 * c.kq)) & 0xFu, hi = (word >> (16 + 4 * c.kq)) & 0xFu;
    return thr == 0 ? 0xFFu : (lo | (hi << 4));
  }
  unsigned o[4];
  philox4x32_10((unsigned)c.grow, (unsigned)((unsigned long long)c.grow >> 32),
                ((unsigned)layer << 16) | ((unsigned)fp << 2) | (unsigned)c.kq, d.stream + c.pass, d.seed_lo, d.seed_hi, o);
  unsigned keep = 0;
#pragma unroll
  for (int w = 0; w < 4; ++w) {
    keep |= ((o[w] & 0xFFFFu) >= thr ? 1u : 0u) << (2 * w);
    keep |= ((o[w] >> 16) >= thr ? 1u : 0u) << (2 * w + 1);
  }
  return keep;
}

// tanh(x) = 1 - 2 / (e^{2x} + 1): v_mul, v_exp_f32, v_add, v_rcp_f32, v_fma -- five VALU issues.
// f32 MFMA and f32 VALU share one datapath on gfx950 (tools/mfma_valu_share.hip: concurrent streams
// take the SUM of their times), so every VALU cycle here is a lost MFMA cycle: keep it minimal.
// Absolute error ~1.5e-7 (exp2 / rcp are 1-ulp approximations); saturates correctly at +-1.
__device__ __forceinline__ float tanh_f32(float x) {
  const float e = __builtin_amdgcn_exp2f(x * 2.8853900817779268f);   // e^{2x}
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}

// ---------------------------------------------------------------------------------------
// weight-slab pipeline: global -> registers -> LDS, two buffers, one barrier per slab
// ---------------------------------------------------------------------------------------
struct ChunkDesc {
  unsigned off;         // float offset of the slab's first element in the flat parameter buffer
  unsigned short ld;    // row stride (floats) of the source matrix
  unsigned char kind;   // 0: forward slab  [rows][32 cols], 16-B chunks XOR-swizzled by (row>>1)&7
                        // 1: backward slab [32 rows][ld cols], column bit 4 XOR-ed with bit 2 of the row
  unsigned char np;     // 16-byte pieces per thread (slab bytes / 4096)
};

struct Pipe {
  const float* params;
  const ChunkDesc* tab;   // in LDS
  char* lds;              // 2 * kChunkBytes
  int n, ci;              // number of slabs in the cycle, slab being computed
  f32x4 regs[8];
  ChunkDesc pending;

  // Branch-free on purpose: with `if (p < np)` around each load hipcc emits s_waitcnt vmcnt(0) before
  // every conditional global_load and the eight slab loads serialise (8 x L2 latency per slab).
  // Slabs smaller than 32 KB simply re-load / re-store their last piece (same address, same data).
  __device__ __forceinline__ void issue(int idx) {
    pending = tab[idx];
    const int tid = threadIdx.x;
    const int fwd_mask = -(int)(pending.kind == 0);
    const int o_f = (tid >> 3) * (int)pending.ld + (tid & 7) * 4, o_b = tid * 4;
    const int s_f = 32 * (int)pending.ld, s_b = 1024;
    const float* g = params + pending.off + (((o_f ^ o_b) & fwd_mask) ^ o_b);
    const int gstride = ((s_f ^ s_b) & fwd_mask) ^ s_b;
    const int last = pending.np - 1;
#pragma unroll
    for (int p = 0; p < 8; ++p) regs[p] = *reinterpret_cast<const f32x4*>(g + (p < last ? p : last) * gstride);
  }
  __device__ __forceinline__ void commit(int buf) {
    const int tid = threadIdx.x;
    char* dst = lds + buf * kChunkBytes;
    const int last = pending.np - 1;
    // forward slab: row = 32 p + tid/8, 16-B chunk (tid & 7) XOR-swizzled by (row >> 1) & 7.
    // backward slab: piece idx = p*256 + tid covers floats [4 idx, 4 idx + 4), row = 4 idx / ld; rows with
    // bit 2 set (kq odd in the reader) get column bit 4 flipped -> conflict-free ds_read_b32.
    const int fwd_off = (tid >> 3) * 128 + (((tid & 7) ^ ((tid >> 4) & 7)) << 4);
    const int sh = 31 - __clz((int)pending.ld) - 2;        // log2(ld / 4)
    const int fwd_mask = -(int)(pending.kind == 0);
#pragma unroll
    for (int p = 0; p < 8; ++p) {
      const int pp = p < last ? p : last;
      const int idx = pp * 256 + tid;
      const int bwd_off = (idx * 16) ^ ((((idx >> sh) >> 2) & 1) << 6);
      const int off = (((fwd_off + pp * 4096) ^ bwd_off) & fwd_mask) ^ bwd_off;   // bit-select: a branch here splits the region
      *reinterpret_cast<f32x4*>(dst + off) = regs[p];
    }
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
    // the slab index cycles with period n; ci itself keeps counting so buffer parity alternates
    issue((ci + 1) % n);
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
// MFMA layer bodies (block = 16 features x 16 rows = one f32x4 per lane)
// ---------------------------------------------------------------------------------------
#define PINN_MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)

template <int NT>
__device__ __forceinline__ void zero_blocks(f32x4 (&v)[NT]) {
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = f32x4{0.f, 0.f, 0.f, 0.f};
}
// accumulators start at the layer's bias (costs the same as zeroing them, saves an add per element later)
template <int NT>
__device__ __forceinline__ void bias_blocks(f32x4 (&v)[NT], const float* __restrict__ bias, int kq) {
#pragma unroll
  for (int t = 0; t < NT; ++t) v[t] = *reinterpret_cast<const f32x4*>(bias + t * 16 + 4 * kq);
}

// out^T[f2][n] += sum_f W[f2][f] h^T[f][n]; one forward slab (32 input features = 2 blocks) per step.
// Output tiles are walked in pairs so consecutive MFMAs hit different accumulators (the
// 16x16x4 form has 40-cycle dependent latency against a 32-cycle issue interval).
template <int NTIN, int NTOUT>
__device__ __forceinline__ void layer_forward(f32x4 (&acc)[NTOUT], const f32x4 (&h)[NTIN], Pipe& pipe, int lane) {
  const int kq = lane >> 4, i = lane & 15;
  const int sw = (i >> 1) & 7;
  const int base = i * 128;
#pragma unroll
  for (int kb = 0; kb < NTIN / 2; ++kb) {
    const char* buf = pipe.cur() + base;
    // flat list of (half, output-tile pair) groups; the A fragments of group g+1 are requested before the
    // eight MFMAs of group g issue, so the ~100-cycle LDS latency sits under 256 cycles of matrix work
    constexpr int kPairs = NTOUT / 2, kGroups = 2 * kPairs;
    f32x4 a0 = *reinterpret_cast<const f32x4*>(buf + (((0 + kq) ^ sw) << 4));
    f32x4 a1 = *reinterpret_cast<const f32x4*>(buf + (((0 + kq) ^ sw) << 4) + 2048);
#pragma unroll
    for (int g = 0; g < kGroups; ++g) {
      const int half = g / kPairs, mt = 2 * (g % kPairs);
      f32x4 n0 = a0, n1 = a1;
      if (g + 1 < kGroups) {
        const int hn = (g + 1) / kPairs, mn = 2 * ((g + 1) % kPairs);
        const int offn = ((4 * hn + kq) ^ sw) << 4;
        n0 = *reinterpret_cast<const f32x4*>(buf + offn + mn * 2048);
        n1 = *reinterpret_cast<const f32x4*>(buf + offn + (mn + 1) * 2048);
      }
      const f32x4 b = h[2 * kb + half];
      acc[mt] = PINN_MFMA16(a0[0], b[0], acc[mt]);
      acc[mt + 1] = PINN_MFMA16(a1[0], b[0], acc[mt + 1]);
      acc[mt] = PINN_MFMA16(a0[1], b[1], acc[mt]);
      acc[mt + 1] = PINN_MFMA16(a1[1], b[1], acc[mt + 1]);
      acc[mt] = PINN_MFMA16(a0[2], b[2], acc[mt]);
      acc[mt + 1] = PINN_MFMA16(a1[2], b[2], acc[mt + 1]);
      acc[mt] = PINN_MFMA16(a0[3], b[3], acc[mt]);
      acc[mt + 1] = PINN_MFMA16(a1[3], b[3], acc[mt + 1]);
      a0 = n0; a1 = n1;
    }
    pipe.advance();
  }
}

// din^T[f][n] += sum_f2 W[f2][f] dpre^T[f2][n]; one backward slab (32 rows of W = 2 dpre blocks) per step
template <int NTK, int NTOUT>
__device__ __forceinline__ void layer_backward(f32x4 (&acc)[NTOUT], const f32x4 (&dpre)[NTK], Pipe& pipe, int lane, int ld) {
  const int kq = lane >> 4, i = lane & 15;
  const int flip = kq & 1;
#pragma unroll
  for (int kb = 0; kb < NTK / 2; ++kb) {
    const float* buf = reinterpret_cast<const float*>(pipe.cur());
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const float* row = buf + (16 * half + 4 * kq) * ld + i;
      const f32x4 b = dpre[2 * kb + half];
#pragma unroll
      for (int mt = 0; mt < NTOUT; mt += 2) {
        const float* c0 = row + ((mt ^ flip) << 4);
        const float* c1 = row + (((mt + 1) ^ flip) << 4);
        acc[mt] = PINN_MFMA16(c0[0], b[0], acc[mt]);
        acc[mt + 1] = PINN_MFMA16(c1[0], b[0], acc[mt + 1]);
        acc[mt] = PINN_MFMA16(c0[ld], b[1], acc[mt]);
        acc[mt + 1] = PINN_MFMA16(c1[ld], b[1], acc[mt + 1]);
        acc[mt] = PINN_MFMA16(c0[2 * ld], b[2], acc[mt]);
        acc[mt + 1] = PINN_MFMA16(c1[2 * ld], b[2], acc[mt + 1]);
        acc[mt] = PINN_MFMA16(c0[3 * ld], b[3], acc[mt]);
        acc[mt + 1] = PINN_MFMA16(c1[3 * ld], b[3], acc[mt + 1]);
      }
    }
    pipe.advance();
  }
}

// input layer: acc = b0 + W0 x^T; W0 [H, 8] straight from global (8 KB, cache resident).
// k-step t contracts inputs {4t + kq}.
template <int NTOUT>
__device__ __forceinline__ void layer_input(f32x4 (&acc)[NTOUT], const float* __restrict__ W0, const float* __restrict__ b0,
                                            const f32x4& xa, const f32x4& xb, int lane) {
  const int kq = lane >> 4, i = lane & 15;
  const float x0 = kq == 0 ? xa[0] : (kq == 1 ? xa[1] : (kq == 2 ? xa[2] : xa[3]));
  const float x1 = kq == 0 ? xb[0] : (kq == 1 ? xb[1] : (kq == 2 ? xb[2] : xb[3]));
#pragma unroll
  for (int mt = 0; mt < NTOUT; ++mt) {
    const float w0 = W0[(mt * 16 + i) * 8 + kq];
    const float w1 = W0[(mt * 16 + i) * 8 + 4 + kq];
    f32x4 c = *reinterpret_cast<const f32x4*>(b0 + mt * 16 + 4 * kq);
    c = PINN_MFMA16(w0, x0, c);
    c = PINN_MFMA16(w1, x1, c);
    acc[mt] = c;
  }
}

// v = dropout(tanh(v)) for the 32-feature group fp (blocks 2fp, 2fp+1); returns its 8 keep bits
// per-layer dropout constants, resolved once outside the unrolled block loop (a select on a
// kernel-argument load inside it turns into a branch per block)
struct LayerDrop {
  unsigned thr;
  float scale;
};
__device__ __forceinline__ LayerDrop layer_drop(const DropDev& d, int mode, int layer) {
  const unsigned on = mode != PINN_DROP_NONE ? 0xFFFFFFFFu : 0u;
  const unsigned t = d.thr[layer];
  const float s = d.scale[layer];
  LayerDrop r;
  r.thr = t & on;
  r.scale = __uint_as_float((__float_as_uint(s) & on) | (0x3F800000u & ~on));
  return r;
}

template <bool kBits>
__device__ __forceinline__ unsigned activate_pair(f32x4& v0, f32x4& v1, const DropDev& d, const RowCtx& c, const LayerDrop ld,
                                                  int layer, int fp) {
  const unsigned thr = ld.thr;
  const float scale = ld.scale;
  const unsigned keep = keep_bits16<kBits>(d, c, thr, layer, fp);
#pragma unroll
  for (int r = 0; r < 4; ++r) {
    const float a0 = tanh_f32(v0[r]);
    const float a1 = tanh_f32(v1[r]);
    v0[r] = ((keep >> r) & 1u) ? a0 * scale : 0.0f;
    v1[r] = ((keep >> (4 + r)) & 1u) ? a1 * scale : 0.0f;
  }
  return keep;
}

// Compute units of the current device, asked once per device (hipGetDeviceProperties costs tens of microseconds: too
// much for the launch path of a 200-us training step at the reference's data sizes).  One process drives one GPU, but
// the cache is keyed by the device id anyway.
static inline int cu_count_cached() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) n = 256;
    cached[dev] = n;
  }
  return cached[dev];
}

// Running moments of the MC-dropout passes (01:1486, np.var with ddof = 0): Welford's update on du = u_t - u_eval.
// The one-pass form E[du^2] - E[du]^2 cancels when the passes nearly coincide (spread << |mean shift|): a row whose
// passes differed by 1e-4 of their common offset lost all digits of e_u.  k = 1-based pass count, inv_k = 1 / k.
__device__ __forceinline__ void welford_update(float& mean, float& m2, float x, float inv_k) {
  const float d = x - mean;
  mean = fmaf(d, inv_k, mean);
  m2 = fmaf(d, x - mean, m2);
}

// tanh only (no dropout module after this layer)
__device__ __forceinline__ void activate_tanh(f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = tanh_f32(v[r]);
}

// partial <w, h> over one block (this lane's 4 features)
__device__ __forceinline__ float block_dot(const f32x4& h, const float* __restrict__ w16, int kq, float s) {
  const f32x4 w = *reinterpret_cast<const f32x4*>(w16 + 4 * kq);
  s = fmaf(w[0], h[0], s);
  s = fmaf(w[1], h[1], s);
  s = fmaf(w[2], h[2], s);
  s = fmaf(w[3], h[3], s);
  return s;
}
// complete a per-lane partial over the four kq lane groups (all lanes get the total)
__device__ __forceinline__ float sum_kq(float s) {
  s += __shfl_xor(s, 16, 64);
  s += __shfl_xor(s, 32, 64);
  return s;
}

// element (feature f, row n) of 16-row tile `t16` of an F-feature tensor lives at ((t16*F + f)*16 + n):
// one block = 4 dword accesses per lane, each wave-instruction four contiguous 64-B segments
__device__ __forceinline__ float* tiled_ptr(float* base, long long t16, int F, int lane) {
  return base + (t16 * F + 4 * (lane >> 4)) * 16 + (lane & 15);
}
// (stash / activation blocks are written once and read by a later phase or kernel: non-temporal, so that they do not push
//  the weights every CU streams out of its XCD's L2 -- see PINN_STASH_ST in pinn_x6_core.h)
__device__ __forceinline__ void store_block(float* __restrict__ p, int ib, const f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) {
#ifdef PINN_ABL_PLAINSTORE
    p[(ib * 16 + r) * 16] = v[r];
#else
    __builtin_nontemporal_store(v[r], p + (ib * 16 + r) * 16);
#endif
  }
}
__device__ __forceinline__ void load_block(const float* __restrict__ p, int ib, f32x4& v) {
#pragma unroll
  for (int r = 0; r < 4; ++r) v[r] = p[(ib * 16 + r) * 16];
}

// workspace views handed from the training entry point to the fp32 / bf16 kernel launchers
struct TrainBuffers {
  void* stash_h;  void* stash_v1; void* stash_v2;     // [.][T16][F][16], fp32 or bf16 elements
  void* dpre_h;   void* dpre_v1;  void* dpre_v2;
  unsigned char* keep;
  float* du; float* dz;
  double* loss_part;
  float* slabs;
  long long t16;
  int n_slices;
  unsigned* amax;     // PINN_PREC_F32X6: bits of max |d pre-activation| over the whole call (non-negative floats order like unsigned
                      // integers: atomicMax, order-independent): the wide nets' fp16 weight-gradient kernels' common scale; the range
                      // record's gradient check (pinn_net_range_status)
  // PINN_PREC_F32X6 on the fused nets (packed stash, pinn_x6_core.h):
  unsigned* emax;     // bits of max over rows of max(|du|, |dz|): the forward kernel's atomicMax, zeroed by the pack kernel before it
  void* rowmeta;      // [t16][256 B]: per tile the rows' scales t_r, the two fp16 parts of du_r * norm_r, dz_r (struct RowMeta)
  int qboost;         // c of t_r = 2^(e_r - E + c): the headroom 8 |h| <= 8 / (1 - p) leaves in fp16
  void* stash_x;      // [t16][2 KB]: the input rows as one packed 32-feature group (8 real features, x / 16): layer 0's weight gradient
};

// kernel arguments of the forward / MC-dropout kernels (fp32 and bf16 variants)
struct FwdArgs {
  const float* params;
  const float* x;
  long long n_rows;
  int H, nh;
  DropDev drop;
  int n_passes;          // MC only
  float* o0;             // forward: u        | MC: pred_mean
  float* o1;             // forward: logvar   | MC: a_u
  float* o2;             //                   | MC: e_u
};

// logvar = log(softplus(z) + 1e-6), softplus with torch's threshold 20 (01:432-434)
__device__ __forceinline__ float softplus_f32(float z) { return z > 20.0f ? z : log1pf(expf(z)); }

// ---------------------------------------------------------------------------------------
// One forward pass of the whole net for this wave's 16 rows -> (u, z); every lane holds them.
// TRAIN: also writes the keep bits (one byte per 32-feature group and lane) and the post-dropout
// activations to the tiled stash.
// ---------------------------------------------------------------------------------------
struct StashPtrs {
  float* h;                // [nh][T16][H][16]
  float* v1;               // [T16][H/2][16]
  float* v2;               // [T16][H/4][16]
  unsigned char* keep;     // [T16][nh*H/32 + H/64][64]
  long long t16_total;
  long long t16;
};

template <int H, bool TRAIN, bool kBits>
__device__ __forceinline__ void forward_pass(const float* __restrict__ P, const float* smallp, const ParamLayout& L, Pipe& pipe, const DropDev& d,
                                             const RowCtx& c, const f32x4& xa, const f32x4& xb, const StashPtrs& st, float& u,
                                             float& z, f32x4 (&v2)[H / 64]) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32;
  const int lane = c.lane, kq = c.kq;
  const SmallLayout S{L.H, L.nh};
  const int n_groups = L.nh * NP + NP / 2;
  unsigned char* keep = TRAIN ? st.keep + (st.t16 * n_groups) * 64 + lane : nullptr;
  f32x4 h[NT];
  layer_input<NT>(h, P + L.w0(), smallp + S.b(0), xa, xb, lane);
#pragma unroll 1
  for (int l = 0; l < L.nh; ++l) {
    // activation of hidden layer l
    const LayerDrop ldr = layer_drop(d, c.mode, l);
    float* sp = TRAIN ? tiled_ptr(st.h + (long long)l * st.t16_total * H * 16, st.t16, H, lane) : nullptr;
#pragma unroll
    for (int fp = 0; fp < NP; ++fp) {
      const unsigned k8 = activate_pair<kBits>(h[2 * fp], h[2 * fp + 1], d, c, ldr, l, fp);
      if (TRAIN) {
        keep[(l * NP + fp) * 64] = (unsigned char)k8;
        store_block(sp, 2 * fp, h[2 * fp]);
        store_block(sp, 2 * fp + 1, h[2 * fp + 1]);
      }
    }
    if (l + 1 < L.nh) {
      f32x4 acc[NT];
      bias_blocks<NT>(acc, smallp + S.b(l + 1), kq);
      layer_forward<NT, NT>(acc, h, pipe, lane);
#pragma unroll
      for (int t = 0; t < NT; ++t) h[t] = acc[t];
    }
  }
  // heads: predict (H -> 1) on the VALU, variance head on the matrix cores
  float up = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t) up = block_dot(h[t], smallp + S.wp() + t * 16, kq, up);
  u = sum_kq(up) + smallp[S.bp()];
  f32x4 v1[NT2];
  bias_blocks<NT2>(v1, smallp + S.bv0(), kq);
  layer_forward<NT, NT2>(v1, h, pipe, lane);
  {
    float* sp = TRAIN ? tiled_ptr(st.v1, st.t16, H / 2, lane) : nullptr;
    const LayerDrop ldr = layer_drop(d, c.mode, L.nh);
#pragma unroll
    for (int fp = 0; fp < NP / 2; ++fp) {
      const unsigned k8 = activate_pair<kBits>(v1[2 * fp], v1[2 * fp + 1], d, c, ldr, L.nh, fp);
      if (TRAIN) {
        keep[(L.nh * NP + fp) * 64] = (unsigned char)k8;
        store_block(sp, 2 * fp, v1[2 * fp]);
        store_block(sp, 2 * fp + 1, v1[2 * fp + 1]);
      }
    }
  }
  bias_blocks<NT4>(v2, smallp + S.bv1(), kq);
  layer_forward<NT2, NT4>(v2, v1, pipe, lane);
  float zp = 0.0f;
  {
    float* sp = TRAIN ? tiled_ptr(st.v2, st.t16, H / 4, lane) : nullptr;
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      activate_tanh(v2[t]);
      zp = block_dot(v2[t], smallp + S.wv2() + t * 16, kq, zp);
      if (TRAIN) store_block(sp, t, v2[t]);
    }
  }
  z = sum_kq(zp) + smallp[S.bv2()];
}

}  // namespace pinn
