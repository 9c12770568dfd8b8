// pinn_x6_core.h -- the fp32-accurate chain on the 16-bit matrix cores: split-operand arithmetic behind one slab machinery.
//
// Why: on gfx950 v_mfma_f32_*_f32 runs at the VALU rate on the vector datapath and blocks every other VALU instruction of
// the SIMD (tools/mfma_valu_share.hip), so the exact-fp32 chain can never hide its tanh / Philox work.  The 16-bit matrix
// cores are 16x faster and independent of the VALU.  Two schemes share the code below (struct X3 / X6 / B1):
//   X3 (every product of PINN_PREC_F32X6): two fp16 parts per operand under exact power-of-two scales, three
//      v_mfma_f32_16x16x32_f16 per product (hi.hi + hi.lo + lo.hi), 22-bit operands, the accuracy of an fp32 matmul;
//      gradients per row under the row's own scale (backward_pass), activations and d pre-activations stashed PACKED as
//      the fragments themselves for the backward and weight-gradient kernels (prep_micro, packed_ptr);
//   X6 (the gradients of the opt-in PINN_PREC_F32X6_G6): three bf16 parts, x = hi + mid + lo exactly, six
//      v_mfma_f32_16x16x32_bf16 per product, fp32's exponent range for every element, fp32 stash.
//
// Structure: one 512-thread workgroup per CU = 8 waves x 16 rows (two waves per SIMD, <= 256 registers each; 4 waves at small
// row counts).  Weights are pre-split and pre-permuted into copies (pack kernel, once per call); one LDS slab = one
// 32-feature K-group of a layer for all output rows = kCopies x [rows][64 B], two slabs in flight, one barrier per slab.
// Activation is LAZY and INTERLEAVED: while slab g multiplies, the raw accumulators of K-group g + 1 are turned into
// their fragments (bias is already in the accumulator; Philox, tanh, dropout, split) in six micro-steps placed between the
// slab's MFMA groups, together with the LDS-DMA pieces of the next slab.  The matrix core arbitrates strictly
// oldest-wave-first, so phase-staggering the two waves of a SIMD serialises instead; with fine interleaving each wave's
// VALU chunk simply runs under the other wave's (and its own) MFMAs.
#pragma once
#include <type_traits>
#include <utility>
#include "pinn_bf16_core.h"

namespace pinn {
namespace x6 {

// compile-time loop: f(std::integral_constant<int, 0>{}) ... f(std::integral_constant<int, N - 1>{}).  The slab code
// indexes register arrays and vector lanes with these; a "runtime" index that the optimiser fails to fold sends
// the whole array to scratch.
template <int I>
using IC = std::integral_constant<int, I>;
template <typename F, int... I>
__device__ __forceinline__ void static_for_impl(F&& f, std::integer_sequence<int, I...>) {
  (f(IC<I>{}), ...);
}
template <int N, typename F>
__device__ __forceinline__ void static_for(F&& f) {
  static_for_impl(f, std::make_integer_sequence<int, N>{});
}

constexpr int kThreadsX = 512;       // 8 waves
constexpr int kTileRowsX = 128;      // rows per workgroup tile
constexpr int kCopyLds = 16384;      // one copy of a slab in LDS: <= 256 rows x 64 B
constexpr int kSlabBytes = 3 * kCopyLds;    // x6: 3 copies (hi, mid, lo)

// packed buffer: three copies (hi, mid, lo) of the bf16 layout of pinn_bf16_core.h (PackLayout), back to back.
// One matrix of it, as the weight stream sees it (wave-uniform; the row stride is a template argument of the users):
struct Mat {
  unsigned off;      // 16-bit-element offset of (row 0, column 0) inside one copy
  int nrb_log;       // log2 of (output rows / 16): 16-row blocks per copy; a slab has kCopies << nrb_log 1-KB pieces
  int kp_log = 0;    // (round 1-2 layout: log2 of the row stride; no longer read)
  int rows_log = -1; // log2 of the matrix's row count where a slab covers only part of the rows (wide nets); -1: the slab's rows are all of them
};
constexpr int clog2(int v) { return v <= 1 ? 0 : 1 + clog2(v >> 1); }

// swizzle of the four 16-B kq chunks of a 64-B row so that ds_read_b128 is conflict-free for the hardware's
// lane groups {0-3,12-15,20-27} ... : chunk' = kq ^ g[(row >> 2) & 3], g = {0, 2, 3, 1}
__device__ __forceinline__ int swz(int row) { return (0x78 >> (2 * ((row >> 2) & 3))) & 3; }   // 2-bit table {0, 2, 3, 1}

typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// Stash stores of the chain: non-temporal.  The 7.7 GB a step writes are read back much later (the d pre-activations only
// by the weight-gradient kernels); written with the default policy they push the 1-MB packed weights that every CU
// streams from its XCD's L2 out of it (chain 4.68 -> 4.28 ms at 1e6 rows; PINN_ABL_PLAINSTORE = the old policy).
#ifdef PINN_ABL_PLAINSTORE
#define PINN_STASH_ST(p, v) (*(p) = (v))
#else
#define PINN_STASH_ST(p, v) __builtin_nontemporal_store((v), (p))
#endif
#ifdef PINN_ABL_NTRING
#define PINN_RING_AUX 2
#else
#define PINN_RING_AUX 0
#endif
#ifdef PINN_X6_STAMP
// diagnostic build only: per-wave cycle sums of the segments of a slab step (first phase, second phase,
// wait + barrier), read back with pinn_x6_debug_read()
__device__ unsigned long long g_x6_stamps[8 * 8];
__device__ __forceinline__ unsigned long long stamp() {
  unsigned long long t;
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t) :: "memory");
  __builtin_amdgcn_sched_barrier(0);
  return t;
}
#define PINN_STAMP(pipe, k) do { const unsigned long long t_ = stamp(); (pipe).seg[k] += t_ - (pipe).last; (pipe).last = t_; } while (0)
#else
#define PINN_STAMP(pipe, k) do { } while (0)
#endif

// Weight stream, global (L2) -> LDS by LDS-DMA (global_load_lds_dwordx4: no staging registers, no ds_write pass).
// One slab = one 32-feature K-group of a matrix for all its output rows = 3 copies x [rows][64 B]; two slabs in LDS:
// while slab k is multiplied, slab k + 1 streams into the other buffer.  One wave-instruction moves one 1-KB piece
// = 16 rows x 64 B of one copy; wave w owns pieces w, w + 8, ... (<= 6).  The LDS image of a piece is lane-linear,
// so the kq-chunk swizzle is applied to the SOURCE address: LDS chunk c of row r holds kq = c ^ swz(r) (an
// involution, the reader applies the same one).  A CU's vector-memory path takes 64 B/clk: issued as one burst
// after the barrier, the 48 pieces of a slab stall all eight waves at issue for ~1000 cycles, so each wave issues
// its pieces one at a time between the MFMAs of its multiply phase (slab_mfma), where the issue slot is free.
template <int NCOPY>
struct PipeT {
  static constexpr int kSlab = NCOPY * kCopyLds;      // bytes of one slab: NCOPY copies (x6: 3 bf16 parts; x3: 2 fp16 parts)
#ifdef PINN_X6_STAMP
  unsigned long long seg[8], last;
#endif
  __amdgpu_buffer_rsrc_t rsrc;   // the packed weights as a raw buffer: copy 0 at byte 0, copies 1, 2 at +copy_bytes
  unsigned copy_bytes;
  char* lds;                 // 2 x kSlab
  int par, wave;             // buffer holding the current slab; wave index (uniform)
  int read_off;              // byte offset of the first row block this wave multiplies (0: all of them; the small-row-count kernels: its quarter)
  unsigned lane_row2, lane_kq8;  // 2 * (lane >> 2), 16 B * ((lane & 3) ^ swz(lane >> 2))

  // A piece is one buffer_load_dwordx4 ... lds: the resource and the piece's byte offset (soffset) are scalar, the
  // per-lane part is one 32-bit VALU op.  (With a flat global address every piece cost two 64-bit multiply-adds and
  // two 64-bit adds on the VALU -- a fifth of the vector issue time of the kernels' main loops.)
  __device__ __forceinline__ void init(const void* packed, unsigned copy_bytes_, int tid) {
    rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(packed), 0, 0x7FFFFFFF, 0x00020000);
    copy_bytes = __builtin_amdgcn_readfirstlane(copy_bytes_);      // (hipcc otherwise keeps it in a VGPR: waterfall loops)
    wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    read_off = 0;
    const int lane = tid & 63;
    lane_row2 = (unsigned)(lane >> 2) << 1;
    lane_kq8 = (unsigned)(((lane & 3) ^ swz(lane >> 2)) << 4);
  }
  // this wave's j-th piece of K-group g of matrix m (row stride 2^KP_LOG) into buffer buf.  Branch-free: a wave whose
  // j-th piece does not exist (small matrices) fetches piece p - n again instead -- same bytes to the same place.
  template <int KP_LOG, int WAVES = 8>
  __device__ __forceinline__ void piece(const Mat& m, int g, int j, int buf) {
    int p = wave + WAVES * j;
    const int n = NCOPY << m.nrb_log;
    p = p < n ? p : p - n;
    asm volatile("" : "+s"(p));   // or hipcc precomputes every piece's offset outside the row loop (register pressure)
    // group-major copies (round 3): [K-group][row][32 elements] -- a slab is ONE contiguous run of rows x 64 B per copy, every
    // 128-B line a piece touches is used whole.  (Row-major copies, rounds 1-2: a piece took 64 B out of each of 16 lines 512 B
    // apart, so a slab pulled twice its bytes through the CU's 64 B/clk L1 fill path -- ~1000 cycles per 32-KB slab, found when the
    // small-row-count kernels' 12-MFMA steps would not go below 1100 cycles.)
    const int rs = m.rows_log >= 0 ? m.rows_log : m.nrb_log + 4;
    const unsigned voff = (lane_row2 << 5) + lane_kq8;                     // bytes, per lane: row (lane >> 2) of the block, 64 B per row
    const int copy = p >> m.nrb_log, rb = p & ((1 << m.nrb_log) - 1);
    const unsigned soff = (unsigned)copy * copy_bytes + 2u * (m.off + (((unsigned)g << rs) << 5) + (unsigned)rb * 512u);
    char* dst = lds + buf * kSlab + copy * kCopyLds + rb * 1024;
#ifndef PINN_ABL_NOPIECE      // (ablation: what a slab step costs without its LDS-DMA instructions; results are garbage)
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
  }
  // The same for a matrix whose shape is a compile-time constant with at least WAVES 16-row blocks per copy (NRB_LOG =
  // log2 of them): piece WAVES * J + wave is (copy, row block) = (static, static + wave), so the scalar offset is the
  // per-wave, per-matrix term mbase = wave_base<KP_LOG>(m) plus compile-time terms: 3 scalar instructions instead of 12.
  template <int KP_LOG>
  __device__ __forceinline__ unsigned wave_base(const Mat& m) const { return 2u * m.off + (unsigned)wave * 1024u; }
  template <int KP_LOG, int WAVES, int NRB_LOG, int J>
  __device__ __forceinline__ void piece_s(unsigned mbase, int g, int buf) {
    constexpr int q = WAVES * J, copy = q >> NRB_LOG, rb0 = q & ((1 << NRB_LOG) - 1);
    static_assert(WAVES <= (1 << NRB_LOG) && copy < NCOPY, "piece_s: shape not separable");
    mbase = __builtin_amdgcn_readfirstlane(mbase);     // (wave-uniform by construction; hipcc does not always see it)
    asm volatile("" : "+s"(mbase));   // (or every piece's offset is precomputed outside the row loop: register pressure)
    const unsigned voff = (lane_row2 << 5) + lane_kq8;
    const unsigned soff = mbase + (unsigned)copy * copy_bytes + ((unsigned)g << (NRB_LOG + 10)) + (unsigned)rb0 * 1024u;      // group stride: (16 << NRB_LOG) rows x 64 B
    char* dst = lds + buf * kSlab + wave * 1024 + copy * kCopyLds + rb0 * 1024;
#ifndef PINN_ABL_NOPIECE
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (lptr_t)dst, 16, voff, soff, 0, 0);
#endif
  }
  // slab 0 of the sequence
  template <int KP_LOG, int WAVES = 8>
  __device__ __forceinline__ void prime(const Mat& first) {
    par = 0;
#pragma unroll
    for (int j = 0; j < 16 * NCOPY / WAVES; ++j)
      if (WAVES * j < (NCOPY << first.nrb_log)) piece<KP_LOG, WAVES>(first, 0, j, 0);
    __syncthreads();           // (drains the DMA: vmcnt(0) + barrier)
  }
  __device__ __forceinline__ const char* cur() const { return lds + par * kSlab + read_off; }
  // end of a slab step: past the barrier every wave is done reading the current slab and the next one is complete.
  // kYoung = vector-memory operations this wave has issued AFTER its last LDS-DMA of the step (the training kernels'
  // stash stores, batched behind the step's last MFMA): vmcnt counts in issue order, so waiting for all but the
  // kYoung youngest retires every DMA of the step and leaves the stores in flight across the barrier.  With
  // __syncthreads() (vmcnt(0)) every step paid the full HBM write latency of its stores: 1.2 of the chain's 4.6 ms.
  template <int kYoung = 0>
  __device__ __forceinline__ void advance() {
#ifdef PINN_ABL_VM0
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#else
    asm volatile("s_waitcnt vmcnt(%0)" :: "n"(kYoung) : "memory");
#endif
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
    par ^= 1;
  }
};
using Pipe6 = PipeT<3>;

// three bf16 fragments of the 8 fp32 values a lane holds in one 32-group: v = hi + mid + lo (exact)
// K order inside a 32-group for the x6 kernels: B-fragment element jj = 2 r + b of lane group kq is feature
// 16 b + 4 kq + r, i.e. the lane's register r of block 2g + b.  One micro-step (register r of both blocks) then
// produces one whole dword of each fragment (v_cvt_pk_bf16_f32), never half of one.
__host__ __device__ inline int pack_col_x6(int q) { return 16 * (q & 1) + 4 * (q >> 3) + ((q & 7) >> 1); }

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
// three bf16 fragments of the 8 fp32 values a lane holds in one 32-group: v = hi + mid + lo (exact); dword r of each
// = (register r of block 0, register r of block 1)
struct Frag3 {
  u32x4 hi, mid, lo;
};
__device__ __forceinline__ unsigned pack_bf16_opaque(float x0, float x1) {
  const bf16x2 h = {(__bf16)x0, (__bf16)x1};
  unsigned p = __builtin_bit_cast(unsigned, h);
  asm volatile("" : "+v"(p));
  return p;
}
template <int R>
__device__ __forceinline__ void split_pair(float x0, float x1, Frag3& f) {
  // one packed conversion per stage; the rounded values come back out of the packed word by a shift / a mask (left to
  // itself hipcc converts every element a second time on its own: 15 instead of 11 instructions per pair)
  const unsigned h = pack_bf16_opaque(x0, x1);
  const float r0 = x0 - __builtin_bit_cast(float, h << 16), r1 = x1 - __builtin_bit_cast(float, h & 0xffff0000u);
  const unsigned m = pack_bf16_opaque(r0, r1);
  const float q0 = r0 - __builtin_bit_cast(float, m << 16), q1 = r1 - __builtin_bit_cast(float, m & 0xffff0000u);
  const bf16x2 l = {(__bf16)q0, (__bf16)q1};
  f.hi[R] = h;
  f.mid[R] = m;
  f.lo[R] = __builtin_bit_cast(unsigned, l);
}
__device__ __forceinline__ Frag3 split3(const f32x4& v0, const f32x4& v1) {
  Frag3 f;
  split_pair<0>(v0[0], v1[0], f);
  split_pair<1>(v0[1], v1[1], f);
  split_pair<2>(v0[2], v1[2], f);
  split_pair<3>(v0[3], v1[3], f);
  return f;
}

struct AFrag3 {
  bf16x8 h, m, l;
};
// A-fragment reads and their waits are written by hand: with an LDS-DMA between the reads hipcc waits lgkmcnt(0)
// before every tile's MFMAs -- for the reads it has just issued too -- and the LDS latency (200+ cycles under load)
// is exposed once per 6 MFMAs (96 cycles).  LDS returns in order, so "at most 3 outstanding" retires the older tile.
template <int OFF>
__device__ __forceinline__ bf16x8 lds_read_b128(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(bf16x8, v);
}
template <int MT>
__device__ __forceinline__ void load_a3(AFrag3& a, unsigned addr) {
  constexpr int kCopy = kCopyLds;
  a.h = lds_read_b128<MT * 1024>(addr);
  a.m = lds_read_b128<kCopy + MT * 1024>(addr);
  a.l = lds_read_b128<2 * kCopy + MT * 1024>(addr);
}
template <int N>
__device__ __forceinline__ void wait_a3(AFrag3& a) {
  asm volatile("s_waitcnt lgkmcnt(%3)" : "+v"(a.h), "+v"(a.m), "+v"(a.l) : "n"(N));
}
__device__ __forceinline__ void mfma6(f32x4& acc, const AFrag3& a, const Frag3& b) {
  f32x4 c = acc;
  const bf16x8 bh = __builtin_bit_cast(bf16x8, b.hi), bm = __builtin_bit_cast(bf16x8, b.mid), bl = __builtin_bit_cast(bf16x8, b.lo);
  c = PINN_MFMA_BF16(a.l, bh, c);
  c = PINN_MFMA_BF16(a.h, bl, c);
  c = PINN_MFMA_BF16(a.m, bm, c);
  c = PINN_MFMA_BF16(a.m, bh, c);
  c = PINN_MFMA_BF16(a.h, bm, c);
  c = PINN_MFMA_BF16(a.h, bh, c);
  acc = c;
}
// ---------------------------------------------------------------------------------------
// x3: TWO fp16 parts, three products (hi.hi + hi.lo + lo.hi) -- the forward passes.
// fp16 has 11 significand bits: x = hi + lo to 2^-22 |x| when lo is a normal fp16, and the dropped lo.lo term is 2^-22 of
// the product: a dot product of 256 such terms is as accurate as torch's fp32 matmul (measured rel. rms 2.5e-7 vs 2.6e-7
// against float64; x6: 8e-8; two bf16 parts: 4e-6), because the fp32 accumulation of the MFMA is then the larger error.
// fp16's narrow exponent is met with exact power-of-two scales: activations x 8 (|h| <= 1.67 / (1 - p)), weights x 64
// (a weight would have to exceed 1023 to overflow), so lo is a normal fp16 wherever it matters (|h| >= 2^-6,
// |w| >= 2^-9; below that the error is an ABSOLUTE 2^-28 / 2^-31 -- fp16 MFMA honours subnormal operands, checked on
// gfx950) and the accumulators carry 512 x the pre-activation: biases are pre-scaled in LDS and the factor leaves in
// the constant of the tanh's exp2.  Gradients (1e-9 .. 1e6 with the 1/N of the loss) do not fit fp16 as they are: the
// backward pass normalises every row by an exact power of two (backward_pass), the weight-gradient kernels scale by row
// (grad_exponent).  Half the MFMAs, 2/3 of the LDS reads and weight DMA of the x6 scheme; the chip is power-bound in these
// kernels, so fewer matrix instructions is what buys time.
// ---------------------------------------------------------------------------------------
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define PINN_MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16((a), (b), (c), 0, 0, 0)
struct Frag2 {
  u32x4 hi, lo;
};
template <int R>
__device__ __forceinline__ void split_pair2(float x0, float x1, Frag2& f) {
  const f16x2 h0 = {(_Float16)x0, (_Float16)x1};
  unsigned hp = __builtin_bit_cast(unsigned, h0);
  asm volatile("" : "+v"(hp));                       // (or hipcc converts each element a second time for the residuals)
  const f16x2 h = __builtin_bit_cast(f16x2, hp);
  const float r0 = x0 - (float)h[0], r1 = x1 - (float)h[1];
  const f16x2 l = {(_Float16)r0, (_Float16)r1};
  f.hi[R] = hp;
  f.lo[R] = __builtin_bit_cast(unsigned, l);
}
struct AFrag2 {
  f16x8 h, l;
};
template <int OFF>
__device__ __forceinline__ f16x8 lds_read_b128_f16(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(f16x8, v);
}

// The two arithmetic schemes behind the slab machinery (slab_pair / slab_mfma / layer_x6 / forward_pass_x6).
struct X6 {
  static constexpr int kCopies = 3;                       // weight copies per slab = LDS reads per tile
  static constexpr float kActScale = 1.0f, kAccScale = 1.0f;
  using Frag = Frag3;
  using AFrag = AFrag3;
  using Pipe = PipeT<3>;
  template <int MT> static __device__ __forceinline__ void load(AFrag& a, unsigned addr) { load_a3<MT>(a, addr); }
  template <int N> static __device__ __forceinline__ void wait(AFrag& a) { wait_a3<N>(a); }
  static __device__ __forceinline__ void mma(f32x4& acc, const AFrag& a, const Frag& b) { mfma6(acc, a, b); }
  template <int R> static __device__ __forceinline__ void split(float x0, float x1, Frag& f) { split_pair<R>(x0, x1, f); }
};
struct X3 {
  static constexpr int kCopies = 2;
  static constexpr float kActScale = 8.0f, kWScale = 64.0f, kAccScale = 512.0f;
  using Frag = Frag2;
  using AFrag = AFrag2;
  using Pipe = PipeT<2>;
  template <int MT> static __device__ __forceinline__ void load(AFrag& a, unsigned addr) {
    a.h = lds_read_b128_f16<MT * 1024>(addr);
    a.l = lds_read_b128_f16<kCopyLds + MT * 1024>(addr);
  }
  template <int N> static __device__ __forceinline__ void wait(AFrag& a) { asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(a.h), "+v"(a.l) : "n"(N)); }
  static __device__ __forceinline__ void mma(f32x4& acc, const AFrag& a, const Frag& b) {
    f32x4 c = acc;
    const f16x8 bh = __builtin_bit_cast(f16x8, b.hi), bl = __builtin_bit_cast(f16x8, b.lo);
    c = PINN_MFMA_F16(a.l, bh, c);
    c = PINN_MFMA_F16(a.h, bl, c);
    c = PINN_MFMA_F16(a.h, bh, c);
    acc = c;
  }
  template <int R> static __device__ __forceinline__ void split(float x0, float x1, Frag& f) { split_pair2<R>(x0, x1, f); }
};

// bf16-mixed (PINN_PREC_BF16 on the wide nets): ONE bf16 part per operand, one MFMA per product, fp32 accumulation -- the
// arithmetic of pinn_bf16_core.h on the slab machinery (the first of the three bf16 weight copies IS bf16(w), round-to-nearest)
struct Frag1 {
  u32x4 hi;
};
struct AFrag1 {
  bf16x8 h;
};
struct B1 {
  static constexpr int kCopies = 1;
  static constexpr float kActScale = 1.0f, kAccScale = 1.0f;
  using Frag = Frag1;
  using AFrag = AFrag1;
  using Pipe = PipeT<1>;
  template <int MT> static __device__ __forceinline__ void load(AFrag& a, unsigned addr) { a.h = lds_read_b128<MT * 1024>(addr); }
  template <int N> static __device__ __forceinline__ void wait(AFrag& a) { asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(a.h) : "n"(N)); }
  static __device__ __forceinline__ void mma(f32x4& acc, const AFrag& a, const Frag& b) {
    acc = PINN_MFMA_BF16(a.h, __builtin_bit_cast(bf16x8, b.hi), acc);
  }
  template <int R> static __device__ __forceinline__ void split(float x0, float x1, Frag& f) {
    const bf16x2 h = {(__bf16)x0, (__bf16)x1};
    f.hi[R] = __builtin_bit_cast(unsigned, h);
  }
};

// One tile pair.  vchunk(IC<c>): VALU chunk c (one per tile) of the next group's preparation; dma(IC<slot>): this
// wave's LDS-DMA piece(s) of the next slab.  Order pinned: MFMAs of a tile, reads of the tile after next, chunk.
// (Measured alternatives, all slower or equal: chunk free to mix with the MFMAs; sched_group_barrier 1 MFMA : 5 VALU;
// waves 4-7 running each chunk before instead of after its tile's MFMAs; whole-phase staggering of the two waves.)
// LDS returns in order: "at most S::kCopies reads outstanding" retires the older tile's fragments.
template <typename S, int MT, int NTOUT, typename V, typename D>
__device__ __forceinline__ void slab_pair(f32x4 (&acc)[NTOUT], const typename S::Frag& b, unsigned addr, typename S::AFrag& a0,
                                          typename S::AFrag& a1, V&& vchunk, D&& dma) {
  S::template wait<S::kCopies>(a0);                 // a1 (issued after a0) may still be in flight
  S::mma(acc[MT], a0, b);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (MT + 2 < NTOUT) S::template load<MT + 2>(a0, addr);
  dma(IC<MT / 2>{});
  vchunk(IC<MT>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (MT + 2 < NTOUT) S::template wait<S::kCopies>(a1); else S::template wait<0>(a1);
  S::mma(acc[MT + 1], a1, b);
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (MT + 3 < NTOUT) S::template load<MT + 3>(a1, addr);
  vchunk(IC<MT + 1>{});
  __builtin_amdgcn_sched_barrier(0);
  if constexpr (MT + 2 < NTOUT) slab_pair<S, MT + 2, NTOUT>(acc, b, addr, a0, a1, vchunk, dma);
}
// the MFMAs of one slab: acc[mt] += A_mt x B over the scheme's parts (x6: terms of order >= 2^-24 dropped).
// Program order is pinned (sched_barrier); two tiles of A fragments in flight.
template <typename S, int NTOUT, typename V, typename D>
__device__ __forceinline__ void slab_mfma(f32x4 (&acc)[NTOUT], const typename S::Frag& b, const char* slab, int lane, V&& vchunk, D&& dma) {
  const int kq = lane >> 4, i = lane & 15;
  const char* base = slab + i * 64 + ((kq ^ swz(i)) << 4);      // rows mt*16 + i: (row >> 2) & 3 == (i >> 2) & 3
  const unsigned addr = (unsigned)(unsigned long long)(lptr_t)base;
  typename S::AFrag a0, a1;
  __builtin_amdgcn_sched_barrier(0);
  S::template load<0>(a0, addr);
  S::template load<1>(a1, addr);
  __builtin_amdgcn_sched_barrier(0);
  slab_pair<S, 0, NTOUT>(acc, b, addr, a0, a1, vchunk, dma);
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

// ---------------------------------------------------------------------------------------
// preparing a 32-feature K-group (two 16-feature blocks v0, v1 of raw pre-activations -> activated in
// place -> Frag3) in six micro-steps: 0, 1 = Philox4x32-10 (five rounds each; one call = the group's eight 16-bit draws);
// 2 .. 5 = register r = k - 2 of both blocks: tanh, dropout, 3-way split (and the predict head's dot).
// ---------------------------------------------------------------------------------------
struct PrepBase {
  unsigned w0, w1, w2, w3;   // Philox counter / output words: one call = eight 16-bit draws = a lane's share of one 32-group
  unsigned keep;             // injected masks (kBits) only
};
template <typename S>
struct PrepT : PrepBase {
  typename S::Frag buf[2];   // fragments of the group in the MFMAs / of the group being prepared, alternating (static parity)
};
using Prep = PrepT<X6>;
// Rounds R0 .. R0 + 4.  The round keys are wave-uniform (seed + round * Weyl constant): they stay on the scalar unit,
// and each counter update is one three-input XOR (v_bitop3_b32, truth table 0x96).
template <int R0>
__device__ __forceinline__ void philox_rounds5(PrepBase& s, unsigned seed_lo, unsigned seed_hi) {
#pragma unroll
  for (int r = R0; r < R0 + 5; ++r) {
    const unsigned long long p0 = (unsigned long long)0xD2511F53u * s.w0;
    const unsigned long long p1 = (unsigned long long)0xCD9E8D57u * s.w2;
    const unsigned n0 = __builtin_amdgcn_bitop3_b32((unsigned)(p1 >> 32), s.w1, seed_lo + (unsigned)r * 0x9E3779B9u, 0x96);
    const unsigned n2 = __builtin_amdgcn_bitop3_b32((unsigned)(p0 >> 32), s.w3, seed_hi + (unsigned)r * 0xBB67AE85u, 0x96);
    s.w1 = (unsigned)p1; s.w3 = (unsigned)p0; s.w0 = n0; s.w2 = n2;
  }
}
// draw of register r of block b of the group: index 4 b + r -> half (index & 1) of word (index >> 1)
template <int B, int R>
__device__ __forceinline__ bool keep_draw(const PrepBase& s, unsigned thr) {
  const unsigned w = B == 0 ? (R < 2 ? s.w0 : s.w1) : (R < 2 ? s.w2 : s.w3);
  return ((R & 1) ? (w >> 16) : (w & 0xFFFFu)) >= thr;
}
// A kept activation that is exactly 0 is stashed as FLT_MIN: the backward pass reads "dropped" off h == 0 (no keep-bit
// stash), and FLT_MIN contributes nothing anywhere (1 - a^2 == 1, products with it underflow).
__device__ __forceinline__ float stash_value(float h, bool kept) {
  return kept ? (h == 0.0f ? 1.17549435e-38f : h) : 0.0f;
}
// tanh(x / acc_scale): `pre` = 2 log2(e) / acc_scale (x3: the accumulators carry 512 x the pre-activation)
constexpr float kTanhPre = 2.8853900817779268f;
__device__ __forceinline__ float tanh_pre(float x, float pre) {
  const float e = __builtin_amdgcn_exp2f(x * pre);
  return fmaf(-2.0f, __builtin_amdgcn_rcpf(e + 1.0f), 1.0f);
}
// wp32: the predict head's 32 weights of this group (LDS) when kDot; the dot is accumulated only if dot_on.
// pre: tanh_pre's constant for the raw values v0, v1; ld.scale is the dropout scale TIMES S::kActScale.
// sp (training only, else nullptr): this lane's slot of the group's first feature in the activation stash.
// FP: the 32-group's index inside its dropout layer (the Philox call index).
// kPack (PINN_PREC_F32X6 on the fused nets): the stash keeps the group as the B fragments just built -- the two fp16 parts of
// S::kActScale x the activation, 16 B per lane and part -- instead of fp32 values: the backward chain rebuilds the activation
// from hi + lo (22 bits, what the forward MFMAs used), and the weight-gradient kernel reads ready operands instead of
// splitting 7.7 GB of fp32 again (round 2: 7.6 VALU instructions per MFMA there).  "Dropped" is still h == 0, so a kept
// activation must never pack to (0, 0): see the kept-zero rule at the tanh below.
template <typename S, bool kBits, bool kDot, int k, int FP, bool kPack = false>
__device__ __forceinline__ void prep_micro(PrepBase& s, f32x4& v0, f32x4& v1, const DropDev& d, const RowCtx& c, const LayerDrop ld, float pre,
                                           int layer, const float* wp32, float& up, bool dot_on, typename S::Frag& out, float* sp = nullptr) {
  constexpr int fp = FP;
  if constexpr (k == 0) {
    if (kBits) {
      const unsigned word = d.bits[((long long)c.pass * c.n_rows + c.lrow) * d.words + layer * d.nb + fp];
      const unsigned lo = (word >> (4 * c.kq)) & 0xFu, hi = (word >> (16 + 4 * c.kq)) & 0xFu;
      s.keep = ld.thr == 0 ? 0xFFu : (lo | (hi << 4));
    } else {
      // the counter passes through an empty volatile asm: otherwise hipcc computes every group's first rounds once,
      // outside the pass loop, and keeps them in registers (spills)
      unsigned kq = (unsigned)c.kq;
      asm volatile("" : "+v"(kq));
      s.w0 = (unsigned)c.grow; s.w1 = (unsigned)((unsigned long long)c.grow >> 32);
      s.w2 = ((unsigned)layer << 16) | ((unsigned)fp << 2) | kq; s.w3 = d.stream + c.pass;
      philox_rounds5<0>(s, d.seed_lo, d.seed_hi);
    }
  } else if constexpr (k == 1) {
    if constexpr (!kBits) philox_rounds5<5>(s, d.seed_lo, d.seed_hi);
  } else if constexpr (k < 6) {
    constexpr int r = k - 2;
    // scale * tanh(x) = scale - 2 scale / (e^{2x} + 1): the dropout / operand scale rides in the tanh's last fma
    const float m2s = -2.0f * ld.scale;
    float a0 = fmaf(m2s, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v0[r] * pre) + 1.0f), ld.scale);
    float a1 = fmaf(m2s, __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v1[r] * pre) + 1.0f), ld.scale);
    if constexpr (kPack) {
      // the backward pass reads "dropped" off h == 0, and this tanh IS exactly 0 for |x| < ~3e-8 (exp2 rounds to 1: about 80
      // kept activations of a 1e6-row step): a kept zero becomes the smallest fp16 subnormal, 2^-24 / kActScale in h
      a0 = a0 == 0.0f ? 0x1p-24f : a0;
      a1 = a1 == 0.0f ? 0x1p-24f : a1;
    }
    const bool k0 = kBits ? ((s.keep >> r) & 1u) != 0 : keep_draw<0, r>(s, ld.thr);
    const bool k1 = kBits ? ((s.keep >> (4 + r)) & 1u) != 0 : keep_draw<1, r>(s, ld.thr);
    const float hs0 = k0 ? a0 : 0.0f;                      // the matrix operand: S::kActScale x the activation
    const float hs1 = k1 ? a1 : 0.0f;
    S::template split<r>(hs0, hs1, out);
    float h0 = hs0, h1 = hs1;                              // the activation itself (stash, predict head)
    if constexpr (S::kActScale != 1.0f) {
      if ((sp && !kPack) || kDot) { h0 = hs0 * (1.0f / S::kActScale); h1 = hs1 * (1.0f / S::kActScale); }
    }
    // training: the registers keep the value the stash will hold (stored by micro-step 6, behind the step's last DMA)
    if constexpr (!kPack) {
      v0[r] = sp ? stash_value(h0, k0) : h0;
      v1[r] = sp ? stash_value(h1, k1) : h1;
    }
    if (kDot) {
      const float t = fmaf(wp32[4 * c.kq + r], h0, wp32[16 + 4 * c.kq + r] * h1);
      up += dot_on ? t : 0.0f;
    }
  } else {
    if (sp) {
      if constexpr (kPack) {      // the fragments as they stand: two 16-B stores
        PINN_STASH_ST(reinterpret_cast<u32x4*>(sp), out.hi);
        PINN_STASH_ST(reinterpret_cast<u32x4*>(sp + 256), out.lo);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          PINN_STASH_ST(sp + r * 16, v0[r]);
          PINN_STASH_ST(sp + (16 + r) * 16, v1[r]);
        }
      }
    }
  }
}

// One layer: NG K-groups; st.buf[P] = the fragments of its group 0 on entry, st.buf[(P + NG) & 1] = those of the next
// layer's group 0 on exit.  prep_in / prep_out get the buffer to fill as their last argument.
// prep_in(g, k): micro-step k of this layer's input group g; prep_out(k): micro-step k of the NEXT layer's group 0,
// whose raw values are this layer's acc[0], acc[1] -- final once the last slab's first tile pair is through, so
// those steps sit in slots >= 1.  KPM / KPN: log2 row stride of this / the next matrix; NPM / NPN: (an upper bound
// of) their 1-KB pieces per slab.
template <typename S, int P, int NG, int NTOUT, int KPM, int KPN, int NPM, int NPN, bool kHasOut, int WAVES = 8, int kStIn = 0, int kStOut = 0,
          int NRBM = -1, typename FI, typename FO>
__device__ __forceinline__ void layer_x6(f32x4 (&acc)[NTOUT], typename S::Pipe& pipe, const Mat& mine, const Mat& next, int lane, PrepT<S>& st,
                                         FI&& prep_in, FO&& prep_out, bool out_on = true) {
  constexpr int kSlots = NTOUT / 2, kPerDma = (16 * S::kCopies / WAVES + kSlots - 1) / kSlots;      // <= 16 kCopies / WAVES pieces per wave and slab
  // VALU chunks: one per tile (NTOUT per slab).  The six micro-steps of the next group go to chunks
  // kFirst + k * (NTOUT - kFirst) / 6; acc[0], acc[1] (the next layer's group 0) are final from chunk 2 on.
  // Micro-step 6 (training: the group's kStIn / kStOut stash stores) is the last thing before the barrier.
  constexpr int kFirst = kHasOut ? 2 : 0, kAvail = NTOUT - kFirst;
  static_assert(kAvail >= 1, "no chunk left for the next layer's group 0");
  // NRBM >= 0: this matrix has the static shape 16 << NRBM rows x (32 NG): the cheap piece addressing (Pipe6::piece_s)
  static_assert(NRBM < 0 || NPM == (S::kCopies << NRBM), "NRBM does not match the piece count");
#ifdef PINN_ABL_GENERIC_PIECE
  constexpr int kNrbm = -1;
#else
  constexpr int kNrbm = NRBM;
#endif
  const unsigned mine_base = kNrbm >= 0 ? pipe.template wave_base<KPM>(mine) : 0u;
  static_for<NG>([&](auto gc) {
    constexpr int g = decltype(gc)::value;
    // the next slab: K-group g + 1 of this matrix, or K-group 0 of the next one
    auto dma = [&](auto slotc) {
      static_for<kPerDma>([&](auto qc) {
        constexpr int j = decltype(slotc)::value * kPerDma + decltype(qc)::value;
        if constexpr (g + 1 < NG) {
          if constexpr (WAVES * j < NPM) {
            if constexpr (kNrbm >= 0) pipe.template piece_s<KPM, WAVES, kNrbm, j>(mine_base, g + 1, pipe.par ^ 1);
            else pipe.template piece<KPM, WAVES>(mine, g + 1, j, pipe.par ^ 1);
          }
        } else {
          if constexpr (WAVES * j < NPN) pipe.template piece<KPN, WAVES>(next, 0, j, pipe.par ^ 1);
        }
      });
    };
    // the next group's fragments
    auto vchunk = [&](auto cc) {
      constexpr int c = decltype(cc)::value;
      static_for<7>([&](auto kc) {
        constexpr int k = decltype(kc)::value;
        if constexpr ((k < 6 && kFirst + k * kAvail / 6 == c) || (k == 6 && c == NTOUT - 1)) {
          if constexpr (g + 1 < NG) prep_in(IC<g + 1>{}, IC<k>{}, st.buf[(P + g + 1) & 1]);
          else if constexpr (kHasOut) prep_out(IC<k>{}, st.buf[(P + g + 1) & 1]);
        }
      });
    };
    slab_mfma<S, NTOUT>(acc, st.buf[(P + g) & 1], pipe.cur(), lane, vchunk, dma);
    PINN_STAMP(pipe, (NTOUT == 16 ? 0 : NTOUT == 8 ? 1 : 2));      // step body; then (below) 4 + type = wait + barrier
    // (out_on: wave-uniform; false when prep_out issues nothing -- the count must never exceed the stores really issued)
    if constexpr (g + 1 < NG) pipe.template advance<kStIn>();
    else if constexpr (kHasOut && kStOut > 0) { if (out_on) pipe.template advance<kStOut>(); else pipe.template advance<0>(); }
    else pipe.template advance<0>();
    PINN_STAMP(pipe, 4 + (NTOUT == 16 ? 0 : NTOUT == 8 ? 1 : 2));
  });
}

// small parameter vectors -> LDS (all threads).  The biases of the matrix layers carry the accumulator scale of scheme S.
// Two phases: every global load of the thread is issued before the first LDS store.  Written as one loop per vector (load,
// store, next vector) the compiler waited out an L2 round trip per loop -- twelve in a row, 5 800 cycles = 2.7 us before a
// kernel's first slab (tools/q_stamps.py), a fifteenth of a training kernel at the reference's row counts.
template <typename S, int kThreads>
__device__ __forceinline__ void fill_small(float* small, float* w0t, const float* __restrict__ params, const ParamLayout& L) {
  const SmallLayout SL{L.H, L.nh};
  const int Hh = L.H, tid = threadIdx.x;
  constexpr int kMaxH = 256, kPer = (kMaxH + kThreads - 1) / kThreads;        // elements of an H-vector per thread
  constexpr int kW0 = (8 * kMaxH + kThreads - 1) / kThreads, kMaxLayers = 8;
  float vb[kMaxLayers][kPer], vwp[kPer], vbv0[kPer], vbv1[kPer], vwv2[kPer], vw0[kW0], vbp = 0.f, vbv2 = 0.f;
#pragma unroll
  for (int l = 0; l < kMaxLayers; ++l)
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const int i = tid + k * kThreads; vb[l][k] = (l < L.nh && i < Hh) ? params[L.b(l) + i] : 0.f; }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kThreads;
    vwp[k] = i < Hh ? params[L.wp() + i] : 0.f;
    vbv0[k] = i < Hh / 2 ? params[L.bv0() + i] : 0.f;
    vbv1[k] = i < Hh / 4 ? params[L.bv1() + i] : 0.f;
    vwv2[k] = i < Hh / 4 ? params[L.wv2() + i] : 0.f;
  }
#pragma unroll
  for (int k = 0; k < kW0; ++k) { const int e = tid + k * kThreads; vw0[k] = e < Hh * 8 ? params[L.w0() + e] : 0.f; }
  if (tid == 0) { vbp = params[L.bp()]; vbv2 = params[L.bv2()]; }
  // ---- stores
#pragma unroll
  for (int l = 0; l < kMaxLayers; ++l)
#pragma unroll
    for (int k = 0; k < kPer; ++k) { const int i = tid + k * kThreads; if (l < L.nh && i < Hh) small[SL.b(l) + i] = l == 0 ? vb[l][k] : vb[l][k] * S::kAccScale; }
#pragma unroll
  for (int k = 0; k < kPer; ++k) {
    const int i = tid + k * kThreads;
    if (i < Hh) small[SL.wp() + i] = vwp[k];
    if (i < Hh / 2) small[SL.bv0() + i] = vbv0[k] * S::kAccScale;
    if (i < Hh / 4) { small[SL.bv1() + i] = vbv1[k] * S::kAccScale; small[SL.wv2() + i] = vwv2[k]; }
  }
#pragma unroll
  for (int k = 0; k < kW0; ++k) { const int e = tid + k * kThreads; if (e < Hh * 8) w0t[(e & 7) * kW0Stride + (e >> 3)] = vw0[k]; }
  if (tid == 0) { small[SL.bp()] = vbp; small[SL.bv2()] = vbv2; }
  __syncthreads();
}

__device__ __forceinline__ int lane_of(const RowCtx& c) { return c.lane; }

// first matrix of the forward slab sequence (the pass after the last one wraps around to it); its row stride is H
template <int H>
__device__ __forceinline__ Mat first_mat(const PackLayout& K) {
  return K.nh > 1 ? Mat{(unsigned)K.w(1), clog2(H / 16)} : Mat{(unsigned)K.wv0(), clog2(H / 32)};
}

// Packed stash (kPack): a 32-feature group of a 16-row tile occupies the same 2 KB the fp32 layout gives it, as
// [part hi / lo][row n][kq][8 x f16]: element 2 r + b of (n, kq) = feature 16 b + 4 kq + r of the group -- the chain's B
// fragment, 16 B per lane and part.  This lane's slot of group 0 (group g: + 512 g floats; part lo: + 256 floats):
__device__ __forceinline__ float* packed_ptr(float* base, long long t16, int F, int lane) {
  return base + t16 * F * 16 + (lane & 15) * 16 + (lane >> 4) * 4;
}

// stash of one 16-row tile: activations and d(pre-activations) as [layer][T16][F][16] fp32 (the fp32 kernels' layout)
struct StashX {
  float* h; float* v1; float* v2;
  float* dh; float* dv1; float* dv2;
  long long t16_total, t16;
  __device__ __forceinline__ float* act(int layer, int H_, int lane) const {      // hidden layer `layer`, feature 0
    return tiled_ptr(h + (long long)layer * t16_total * H_ * 16, t16, H_, lane);
  }
  __device__ __forceinline__ float* dact(int layer, int H_, int lane) const {
    return tiled_ptr(dh + (long long)layer * t16_total * H_ * 16, t16, H_, lane);
  }
  __device__ __forceinline__ float* actp(int layer, int H_, int lane) const {     // packed forms
    return packed_ptr(h + (long long)layer * t16_total * H_ * 16, t16, H_, lane);
  }
  __device__ __forceinline__ float* dactp(int layer, int H_, int lane) const {
    return packed_ptr(dh + (long long)layer * t16_total * H_ * 16, t16, H_, lane);
  }
};

// One forward pass for this wave's 16 rows in arithmetic scheme S (X3: the kernels' forward passes; X6 kept for
// measurement).  Returns (u, z); TRAIN: the activations go to the stash (v2 = the tanh'ed last variance blocks too).  The
// weight stream arrives and leaves positioned on the forward sequence's first matrix.  `smallp` holds the biases of the
// matrix layers (b_1.., bv_0, bv_1) TIMES S::kAccScale (the kernels scale them when they fill the LDS), b_0 and the head
// vectors as they are.
template <typename S, int H, bool kBits, bool TRAIN = false, int WAVES = 8, bool kPack = false>
__device__ __forceinline__ void forward_pass(const float* w0t, const float* smallp, const ParamLayout& L, typename S::Pipe& pipe,
                                             const DropDev& d, const RowCtx& c, const f32x4& xa, const f32x4& xb, float& u, float& z,
                                             const StashX* sx = nullptr) {
  static_assert(!kPack || (TRAIN && S::kCopies == 2), "the packed stash holds scheme X3's fragments");
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32, NC = S::kCopies;
  constexpr int kSt = TRAIN ? (kPack ? 2 : 8) : 0;      // stash stores per prepared group (behind the step's last DMA)
  auto act_ptr = [&](int layer) -> float* { return kPack ? sx->actp(layer, H, lane_of(c)) : sx->act(layer, H, lane_of(c)); };
  auto v1_ptr = [&]() -> float* { return kPack ? packed_ptr(sx->v1, sx->t16, H / 2, lane_of(c)) : tiled_ptr(sx->v1, sx->t16, H / 2, lane_of(c)); };
  using Frag = typename S::Frag;
  const int lane = c.lane, kq = c.kq;
  const SmallLayout SL{L.H, L.nh};
  const PackLayout K{L.H, L.nh};
  constexpr int KPW = clog2(H), KPV1 = clog2((H / 2 + 63) & ~63);          // row strides: [H][H], [H/2][H]; [H/4][H/2]
  const Mat m_v0{(unsigned)K.wv0(), clog2(H / 32)}, m_v1{(unsigned)K.wv1(), clog2(H / 64)};
  const float* wp = smallp + SL.wp();
  const int ll = L.nh - 1;
  constexpr float kPre0 = kTanhPre, kPreS = kTanhPre / S::kAccScale;     // tanh of an unscaled / a scaled accumulator
  auto drop_of = [&](int layer) { LayerDrop ld = layer_drop(d, c.mode, layer); ld.scale *= S::kActScale; return ld; };
  float up = 0.0f;
  PrepT<S> st;
  f32x4 h[NT];
  layer_input_lds<NT>(h, w0t, smallp + SL.b(0), xa, xb, lane);      // 8 -> H in exact fp32 (K = 8)
  {   // group 0 of the first matrix layer's input: nothing to hide it under
    const LayerDrop ld0 = drop_of(0);
    float* sp = TRAIN ? act_ptr(0) : nullptr;
    static_for<7>([&](auto kc) { prep_micro<S, kBits, true, decltype(kc)::value, 0, kPack>(st, h[0], h[1], d, c, ld0, kPre0, 0, wp, up, ll == 0, st.buf[0], sp); });
  }
#pragma unroll 1
  for (int l = 1; l < L.nh; ++l) {
    f32x4 acc[NT];
    bias_blocks<NT>(acc, smallp + SL.b(l), kq);
    const LayerDrop ld_in = drop_of(l - 1), ld_out = drop_of(l);
    const float pre_in = l == 1 ? kPre0 : kPreS;                    // layer 0's output comes from the exact-fp32 input layer
    const bool last = l == ll;
    const Mat mine{(unsigned)K.w(l), clog2(H / 16)}, next = last ? m_v0 : Mat{(unsigned)K.w(l + 1), clog2(H / 16)};
    float* sp_in = TRAIN ? act_ptr(l - 1) : nullptr;
    float* sp_out = TRAIN ? act_ptr(l) : nullptr;
    static_assert(NP % 2 == 0 && (NP / 2) % 2 == 0, "the forward layers keep the fragment buffer parity");
    layer_x6<S, 0, NP, NT, KPW, KPW, NC * H / 16, NC * H / 16, true, WAVES, kSt, kSt, clog2(H / 16)>(
        acc, pipe, mine, next, lane, st,
        [&](auto gc, auto kc, Frag& out) {
          constexpr int g = decltype(gc)::value;
          prep_micro<S, kBits, false, decltype(kc)::value, g, kPack>(st, h[2 * g], h[2 * g + 1], d, c, ld_in, pre_in, l - 1, wp, up, false, out,
                                                                    TRAIN ? sp_in + 32 * g * 16 : nullptr);
        },
        [&](auto kc, Frag& out) {
          prep_micro<S, kBits, true, decltype(kc)::value, 0, kPack>(st, acc[0], acc[1], d, c, ld_out, kPreS, l, wp, up, last, out, sp_out);
        });
#pragma unroll
    for (int t = 0; t < NT; ++t) h[t] = acc[t];
  }
  // variance head, first layer: its input is the last hidden layer (group 0 is prepared already); predict head on the fly
  f32x4 v1[NT2];
  bias_blocks<NT2>(v1, smallp + SL.bv0(), kq);
  {
    const LayerDrop ld_in = drop_of(ll), ld_out = drop_of(L.nh);
    const float pre_in = ll == 0 ? kPre0 : kPreS;
    float* sp_in = TRAIN ? act_ptr(ll) : nullptr;
    float* sp_out = TRAIN ? v1_ptr() : nullptr;
    layer_x6<S, 0, NP, NT2, KPW, KPV1, NC * H / 32, NC * H / 64, true, WAVES, kSt, kSt, (WAVES <= H / 32 ? clog2(H / 32) : -1)>(
        v1, pipe, m_v0, m_v1, lane, st,
        [&](auto gc, auto kc, Frag& out) {
          constexpr int g = decltype(gc)::value;
          prep_micro<S, kBits, true, decltype(kc)::value, g, kPack>(st, h[2 * g], h[2 * g + 1], d, c, ld_in, pre_in, ll, wp + 32 * g, up, true, out,
                                                                   TRAIN ? sp_in + 32 * g * 16 : nullptr);
        },
        [&](auto kc, Frag& out) {
          prep_micro<S, kBits, false, decltype(kc)::value, 0, kPack>(st, v1[0], v1[1], d, c, ld_out, kPreS, L.nh, wp, up, false, out, sp_out);
        });
  }
  u = sum_kq(up) + smallp[SL.bp()];
  f32x4 v2[NT4];
  bias_blocks<NT4>(v2, smallp + SL.bv1(), kq);
  {
    const LayerDrop ld_in = drop_of(L.nh);
    float* sp_in = TRAIN ? v1_ptr() : nullptr;
    auto prep_in = [&](auto gc, auto kc, Frag& out) {
      constexpr int g = decltype(gc)::value;
      prep_micro<S, kBits, false, decltype(kc)::value, g, kPack>(st, v1[2 * g], v1[2 * g + 1], d, c, ld_in, kPreS, L.nh, wp, up, false, out,
                                                                TRAIN ? sp_in + 32 * g * 16 : nullptr);
    };
    layer_x6<S, 0, NP / 2, NT4, KPV1, KPW, NC * H / 64, NC * H / 16, false, WAVES, kSt, 0>(v2, pipe, m_v1, first_mat<H>(K), lane, st, prep_in,
                                                                                          [&](auto, Frag&) {});
  }
  float zp = 0.0f;
  float* sp2 = TRAIN ? tiled_ptr(sx->v2, sx->t16, H / 4, lane) : nullptr;
#pragma unroll
  for (int t = 0; t < NT4; ++t) {
#pragma unroll
    for (int r = 0; r < 4; ++r) v2[t][r] = tanh_pre(v2[t][r], kPreS);
    zp = block_dot(v2[t], smallp + SL.wv2() + t * 16, kq, zp);
    if (TRAIN) store_block(sp2, t, v2[t]);
  }
  z = sum_kq(zp) + smallp[SL.bv2()];
}

// ---------------------------------------------------------------------------------------
// backward pass
// ---------------------------------------------------------------------------------------
// This wave's LDS copies of stash blocks (32 features x 16 rows fp32 = 2 KB, the global layout as it is): the block
// of the group that will be prepared in the NEXT slab step streams in by LDS-DMA during this one (retired by the
// step's barrier), so the backward chunks never wait on a global load.
struct StashRing {
  char* lds;      // 2 x 2048 B of this wave
  int lane;
  __device__ __forceinline__ void fetch(const float* block, int buf) const {
    const char* src = reinterpret_cast<const char*>(block) + lane * 16;
    __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(lds + buf * 2048), 16, 0, PINN_RING_AUX);
    __builtin_amdgcn_global_load_lds((gptr_t)(src + 1024), (lptr_t)(lds + buf * 2048 + 1024), 16, 0, PINN_RING_AUX);
  }
  // feature 16 b + 4 kq + r of the block, this lane's row
  __device__ __forceinline__ float read(int buf, int b, int r) const {
    return *reinterpret_cast<const float*>(lds + buf * 2048 + ((16 * b + 4 * (lane >> 4) + r) * 16 + (lane & 15)) * 4);
  }
  // packed block: this lane's fragment of part `part` (0 = hi, 1 = lo)
  __device__ __forceinline__ u32x4 read_frag(int buf, int part) const {
    return *reinterpret_cast<const u32x4*>(lds + buf * 2048 + part * 1024 + (lane & 15) * 64 + (lane >> 4) * 16);
  }
};

// the two fp16 parts of a packed stash group, as this lane holds them (dword r = (block 0, block 1) of register r)
struct StashFrag {
  u32x4 hi, lo;
};
// S::kActScale x the activation of block b, register r: hi + lo (exact in fp32)
template <int B, int R>
__device__ __forceinline__ float unpack_act(const StashFrag& f) {
  const unsigned hw = f.hi[R], lw = f.lo[R];      // (scalars first: __builtin_bit_cast of a vector element reads element 0, hipcc 7.2)
  const f16x2 h = __builtin_bit_cast(f16x2, hw), l = __builtin_bit_cast(f16x2, lw);
  return (float)h[B] + (float)l[B];
}

// Row scale of the packed weight gradients.  The packed d pre-activations are the backward chain's operands: the row's
// gradients times norm_r = 2^(4 - e_r), e_r the exponent of max(|du|, |dz|).  dW = sum_r d pre_r (x) h_r needs the true ones,
// so the weight-gradient kernel multiplies the OTHER operand's row by t_r = 2^(e_r - E + c) (one v_pk_mul_f16 per dword),
// E = the call's largest e_r (from TrainBuffers::emax, measured by the forward kernel), c = the headroom 8 |h| leaves in
// fp16; the sums come out times 2^(4 - E + c).  Rows without gradient (all-zero or padding) take e_r = E.
__device__ __forceinline__ int grad_exponent(float mx, int e_if_zero) {
  int e = 0;
  (void)frexpf(mx, &e);
  return (mx > 0.0f && mx < __builtin_inff()) ? (e < -120 ? -120 : e) : e_if_zero;
}

// micro-step k of a backward group: raw d0, d1 = (acc_scale x) d loss / d h (two 16-feature blocks) -> d pre-activation in place,
// split for the next matrix (k = 2 .. 5) and stashed for the weight-gradient kernels (dsp; k = 6).  h = post-dropout activation from
// the stash copy in LDS: dropped <=> h == 0, a = h / drop_scale, d pre = d h * gscale * (1 - a^2) with gscale = the dropout
// scale over the accumulator scale of the matrix that produced d0, d1.  The values in flight are the row's gradients times
// its normalisation (scheme X3, see backward_pass); the stash gets them back in true units: x unnorm.
template <typename S, int k>
__device__ __forceinline__ void bprep_micro(typename S::Frag& out, f32x4& d0, f32x4& d1, const StashRing& ring, int buf, float* dsp, float gscale,
                                            float inv_scale, float unnorm, float& mx, StashFrag& sf) {
  constexpr bool kPack = S::kCopies == 2;       // scheme X3 (PINN_PREC_F32X6): packed stash, see prep_micro
  if constexpr (k == 1) {
    if constexpr (kPack) { sf.hi = ring.read_frag(buf, 0); sf.lo = ring.read_frag(buf, 1); }
  } else if constexpr (k >= 2 && k < 6) {
    constexpr int r = k - 2;
    // packed: inv_scale = 1 / (dropout scale x S::kActScale), the stash holds kActScale x h
    const float h0 = kPack ? unpack_act<0, r>(sf) : ring.read(buf, 0, r), h1 = kPack ? unpack_act<1, r>(sf) : ring.read(buf, 1, r);
    const float a0 = h0 * inv_scale, a1 = h1 * inv_scale;
    const float g0 = d0[r] * (gscale * (1.0f - a0 * a0)), g1 = d1[r] * (gscale * (1.0f - a1 * a1));
    const float p0 = h0 != 0.0f ? g0 : 0.0f, p1 = h1 != 0.0f ? g1 : 0.0f;
    d0[r] = p0; d1[r] = p1;
    S::template split<r>(p0, p1, out);
  } else if constexpr (k == 6) {      // the group's stores, behind the step's last DMA (PipeT::advance<kYoung>)
    if constexpr (kPack) {            // the operands just built, in the row's normalised units (the weight-gradient kernel scales by row)
      PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp), out.hi);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp + 256), out.lo);
#pragma unroll
      for (int r = 0; r < 4; ++r) mx = fmaxf(fmaxf(mx, fabsf(d0[r])), fabsf(d1[r]));      // (normalised units; v_max3_f32)
    } else {
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        PINN_STASH_ST(dsp + r * 16, S::kActScale != 1.0f ? d0[r] * unnorm : d0[r]);
        PINN_STASH_ST(dsp + (16 + r) * 16, S::kActScale != 1.0f ? d1[r] * unnorm : d1[r]);
        if constexpr (S::kActScale != 1.0f) mx = fmaxf(fmaxf(mx, fabsf(d0[r])), fabsf(d1[r]));
      }
    }
  }
}

// Backward chain of this wave's 16 rows in scheme S: d pre-activations of every layer to the stash.  du, dz = d loss /
// d (u, z); the tanh'ed last variance blocks come from the stash (the forward pass is a kernel of its own).  The weight
// stream arrives positioned on Wv1^T and leaves there (the next tile's backward pass).
// X3: the gradients (1 / N of the loss in front, precisions up to 1e6: 1e-9 .. 1e6) do not fit fp16 as they are -- but a
// row is one COLUMN of every product W^T d, so it may carry its own scale: each row's (du, dz) is normalised by an exact
// power of two to max(|du|, |dz|) in [8, 16) (`norm`), the whole chain of that row runs on normalised values (bounded by
// the weights' row sums, far inside fp16's range), and what goes to the stash is multiplied by 1 / norm again (exact).
// What the packed weight-gradient kernels need besides the stash (scheme X3 only): E and the headroom c of the row scale
// t_r (grad_exponent above), and this tile's 256-B record: fp16 [0..15] t_r, [16..31] / [32..47] the two fp16 parts of
// du_r * norm_r (the predict head's weight gradient is a dot product with h like any other row of d pre); fp32 [32..47]
// (bytes 128 ..) dz_r for the variance head's last weight, whose other operand stays fp32.
struct RowMeta {
  _Float16* rec;      // this tile's record (128 x f16), or nullptr
  int E, c;
};
template <typename S, int H, int WAVES = 8>
__device__ __forceinline__ void backward_pass(const float* smallp, const ParamLayout& L, typename S::Pipe& pipe, const DropDev& d, int mode,
                                              const StashX& sx, const StashRing& ring, int lane, float du, float dz, float& amax,
                                              const RowMeta meta = RowMeta{nullptr, 4, 0}) {
  constexpr int NT = H / 16, NT2 = H / 32, NT4 = H / 64, NP = H / 32, NC = S::kCopies;
  constexpr int NG1 = (H / 4) / 32 > 0 ? (H / 4) / 32 : 1;                 // K-groups of Wv1^T (K = H/4)
  constexpr int KPW = clog2(H), KPT0 = clog2((H / 2 + 63) & ~63), KPT1 = clog2((H / 4 + 63) & ~63);
  constexpr bool kPack = NC == 2;                                          // scheme X3: packed stash (prep_micro)
  constexpr int kSt = kPack ? 2 : 8;                                       // stash stores per prepared group
  using Frag = typename S::Frag;
  const int kq = lane >> 4;
  const SmallLayout SL{L.H, L.nh};
  const PackLayout K{L.H, L.nh};
  const int nh = L.nh;
  const int n_blocks = NP / 2 + (nh - 1) * NP;                             // stash blocks that feed a matrix, in order
  // i-th stash block in backward order: v1 groups 0 .. NP/2-1, then hidden layers nh-1 .. 1, groups 0 .. NP-1
  auto block_ptr = [&](int i) -> const float* {
    if (i < NP / 2) return sx.v1 + (sx.t16 * (H / 2) + 32 * i) * 16;
    const int j = i - NP / 2, layer = nh - 1 - j / NP, g = j % NP;
    return sx.h + ((long long)layer * sx.t16_total * H + sx.t16 * H + 32 * g) * 16;
  };
  auto fetch_block = [&](int i) {
    if (i < n_blocks) ring.fetch(block_ptr(i), i & 1);
  };
  fetch_block(0);

  // per-row normalisation (X3): norm = 2^(4 - e) with max(|du|, |dz|) = m 2^e, m in [0.5, 1)
  float unnorm = 1.0f;
  float mx = 0.0f;        // X3: max |d pre-activation| this lane stashes for this tile, in the row's normalised units
  if constexpr (S::kActScale != 1.0f) {
    const int e = grad_exponent(fmaxf(fabsf(du), fabsf(dz)), kPack ? meta.E : 4);      // (all-zero row, padding: no gradient to scale)
    const float norm = ldexpf(1.0f, 4 - e);
    unnorm = ldexpf(1.0f, e - 4);
    if constexpr (kPack) {
      if (meta.rec && lane < 16) reinterpret_cast<float*>(meta.rec)[32 + lane] = dz;
    }
    du *= norm; dz *= norm;
    if constexpr (kPack) {
      if (meta.rec && lane < 16) {
        const _Float16 dh16 = (_Float16)du;
        meta.rec[lane] = (_Float16)ldexpf(1.0f, e - meta.E + meta.c);      // t_r <= 2^c; a power of two down to 2^-24, 0 below
        meta.rec[16 + lane] = dh16;
        meta.rec[32 + lane] = (_Float16)(du - (float)dh16);
      }
    }
  }
  constexpr float kInvW = 1.0f / (S::kAccScale / S::kActScale);            // 1 / weight scale: accumulators carry kWScale x W^T d
  constexpr float kInvA = kPack ? 1.0f / S::kActScale : 1.0f;              // the packed stash holds kActScale x h

  PrepT<S> st;
  StashFrag sf;
  constexpr int P1 = NG1 & 1;      // fragment buffer parity after Wv1^T (its group count is odd for H = 128)
  // ---- d pre_v2 = wv2 * dz * (1 - v2^2): the B operand of Wv1^T, all in registers
  f32x4 v2[NT4];
  float* dsp2 = kPack ? packed_ptr(sx.dv2, sx.t16, H / 4, lane) : tiled_ptr(sx.dv2, sx.t16, H / 4, lane);
  {
    const float* vp = tiled_ptr(sx.v2, sx.t16, H / 4, lane);
#pragma unroll
    for (int t = 0; t < NT4; ++t) load_block(vp, t, v2[t]);
#pragma unroll
    for (int t = 0; t < NT4; ++t) {
      const f32x4 w = *reinterpret_cast<const f32x4*>(smallp + SL.wv2() + t * 16 + 4 * kq);
      f32x4 true_units;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v2[t][r] = w[r] * dz * (1.0f - v2[t][r] * v2[t][r]);
        true_units[r] = v2[t][r] * unnorm;
        if constexpr (S::kActScale != 1.0f) mx = fmaxf(mx, fabsf(v2[t][r]));
      }
      if constexpr (!kPack) store_block(dsp2, t, true_units);
    }
    const f32x4 zero = {0.f, 0.f, 0.f, 0.f};
    const f32x4& second = NT4 >= 2 ? v2[NT4 >= 2 ? 1 : 0] : zero;           // H = 128: K = 32 of a padded 64, one real block
    static_for<4>([&](auto rc) { constexpr int r = decltype(rc)::value; S::template split<r>(v2[0][r], second[r], st.buf[0]); });
    if constexpr (kPack) {      // group 0 of d pre_v2, packed (further groups: with their split, below)
      PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp2), st.buf[0].hi);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp2 + 256), st.buf[0].lo);
    }
  }
  const Mat m_t1{(unsigned)K.wv1t(), clog2(H / 32)}, m_t0{(unsigned)K.wv0t(), clog2(H / 16)};
  const Mat m_again{m_t1.off, m_t1.nrb_log, KPT1};                          // the next tile's first matrix (run-time row stride)
  auto dv1_ptr = [&]() -> float* { return kPack ? packed_ptr(sx.dv1, sx.t16, H / 2, lane) : tiled_ptr(sx.dv1, sx.t16, H / 2, lane); };
  auto dact_ptr = [&](int layer) -> float* { return kPack ? sx.dactp(layer, H, lane) : sx.dact(layer, H, lane); };

  // ---- d h_v1 = Wv1^T d pre_v2; lazily -> d pre_v1 (stash block i = its group)
  f32x4 dpv1[NT2];
  zero_blocks<NT2>(dpv1);
  {
    const LayerDrop ldv = layer_drop(d, mode, nh);
    const float gscale = ldv.scale * kInvW, inv_scale = kInvA / ldv.scale;
    float* dsp = dv1_ptr();
    layer_x6<S, 0, NG1, NT2, KPT1, KPT0, NC * H / 32, NC * H / 16, true, WAVES, (kPack ? 2 : 0), kSt, (WAVES <= H / 32 ? clog2(H / 32) : -1)>(
        dpv1, pipe, m_t1, m_t0, lane, st,
        [&](auto gc, auto kc, Frag& out) {
          constexpr int g = decltype(gc)::value, k = decltype(kc)::value;
          if constexpr (k >= 2 && k < 6 && 2 * g + 1 < NT4) S::template split<k - 2>(v2[2 * g][k - 2], v2[2 * g + 1][k - 2], out);
          if constexpr (kPack && k == 6 && 2 * g + 1 < NT4) {
            PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp2 + 512 * g), out.hi);
            PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp2 + 512 * g + 256), out.lo);
          }
        },
        [&](auto kc, Frag& out) {
          constexpr int k = decltype(kc)::value;
          if constexpr (k == 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // block 0 was requested at the top of this pass
            fetch_block(1);
          }
          bprep_micro<S, k>(out, dpv1[0], dpv1[1], ring, 0, dsp, gscale, inv_scale, unnorm, mx, sf);
        });
  }

  // ---- d h_last = w_p du + Wv0^T d pre_v1; lazily -> d pre of the last hidden layer
  f32x4 dh[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const f32x4 w = *reinterpret_cast<const f32x4*>(smallp + SL.wp() + t * 16 + 4 * kq);
    dh[t] = w * (du * (S::kAccScale / S::kActScale));                       // the accumulator carries the weight scale
  }
  {
    const LayerDrop ldv = layer_drop(d, mode, nh), ldh = layer_drop(d, mode, nh - 1);
    const float gscale = ldv.scale * kInvW, inv_scale = kInvA / ldv.scale, gscale_o = ldh.scale * kInvW, inv_scale_o = kInvA / ldh.scale;
    float* dsp = dv1_ptr();
    float* dsp_o = dact_ptr(nh - 1);
    const Mat next = nh > 1 ? Mat{(unsigned)K.wt(nh - 1), clog2(H / 16), KPW} : m_again;
    layer_x6<S, P1, NP / 2, NT, KPT0, -1, NC * H / 16, NC * H / 16, true, WAVES, kSt, kSt, clog2(H / 16)>(
        dh, pipe, m_t0, next, lane, st,
        [&](auto gc, auto kc, Frag& out) {
          constexpr int g = decltype(gc)::value, k = decltype(kc)::value;
          if constexpr (k == 0) fetch_block(g + 1);
          bprep_micro<S, k>(out, dpv1[2 * g], dpv1[2 * g + 1], ring, g & 1, dsp + 32 * g * 16, gscale, inv_scale, unnorm, mx, sf);
        },
        [&](auto kc, Frag& out) {
          constexpr int k = decltype(kc)::value;
          if (nh > 1) {
            if constexpr (k == 0) fetch_block(NP / 2 + 1);
            bprep_micro<S, k>(out, dh[0], dh[1], ring, (NP / 2) & 1, dsp_o, gscale_o, inv_scale_o, unnorm, mx, sf);
          }
        },
        nh > 1);
  }

  // ---- hidden layers nh-1 .. 1: d h_{l-1} = W_l^T d pre_l
#pragma unroll 1
  for (int l = nh - 1; l >= 1; --l) {
    f32x4 acc[NT];
    zero_blocks<NT>(acc);
    const LayerDrop ld_in = layer_drop(d, mode, l), ld_out = layer_drop(d, mode, l - 1);
    const float gscale = ld_in.scale * kInvW, inv_scale = kInvA / ld_in.scale, gscale_o = ld_out.scale * kInvW, inv_scale_o = kInvA / ld_out.scale;
    float* dsp = dact_ptr(l);
    float* dsp_o = dact_ptr(l - 1);
    const int base = NP / 2 + (nh - 1 - l) * NP;           // stash block index of this layer's group 0
    const Mat mine{(unsigned)K.wt(l), clog2(H / 16)}, next = l > 1 ? Mat{(unsigned)K.wt(l - 1), clog2(H / 16), KPW} : m_again;
    layer_x6<S, P1, NP, NT, KPW, -1, NC * H / 16, NC * H / 16, true, WAVES, kSt, kSt, clog2(H / 16)>(
        acc, pipe, mine, next, lane, st,
        [&](auto gc, auto kc, Frag& out) {
          constexpr int g = decltype(gc)::value, k = decltype(kc)::value;
          if constexpr (k == 0) fetch_block(base + g + 1);
          bprep_micro<S, k>(out, dh[2 * g], dh[2 * g + 1], ring, (base + g) & 1, dsp + 32 * g * 16, gscale, inv_scale, unnorm, mx, sf);
        },
        [&](auto kc, Frag& out) {
          constexpr int k = decltype(kc)::value;
          if (l > 1) {
            if constexpr (k == 0) fetch_block(base + NP + 1);
            bprep_micro<S, k>(out, acc[0], acc[1], ring, (base + NP) & 1, dsp_o, gscale_o, inv_scale_o, unnorm, mx, sf);
          }
        },
        l > 1);
#pragma unroll
    for (int t = 0; t < NT; ++t) dh[t] = acc[t];
  }

  // ---- layer 0: d pre_0 = d h_0 * tanh' (no matrix follows: plain loads, all issued before the arithmetic).  Packed like the
  //      others in scheme X3 (its weight gradient reads the packed input rows, stash_x); fp32 in true units otherwise
  {
    const LayerDrop ld0 = layer_drop(d, mode, 0);
    const float gscale = ld0.scale * kInvW * (kPack ? 1.0f : unnorm), inv_scale = kInvA / ld0.scale;
    auto one = [&](float h, float dv) -> float {
      const float a0 = h * inv_scale;
      const float g0 = dv * (gscale * (1.0f - a0 * a0));
      return h != 0.0f ? g0 : 0.0f;
    };
    if constexpr (kPack) {
      const float* hp = sx.actp(0, H, lane);
      float* dsp = sx.dactp(0, H, lane);
      StashFrag fr[NP];
#pragma unroll
      for (int g = 0; g < NP; ++g) {
        fr[g].hi = *reinterpret_cast<const u32x4*>(hp + 512 * g);
        fr[g].lo = *reinterpret_cast<const u32x4*>(hp + 512 * g + 256);
      }
      static_for<NP>([&](auto gc) {
        constexpr int g = decltype(gc)::value;
        Frag out;
        static_for<4>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          const float p0 = one(unpack_act<0, r>(fr[g]), dh[2 * g][r]), p1 = one(unpack_act<1, r>(fr[g]), dh[2 * g + 1][r]);
          mx = fmaxf(fmaxf(mx, fabsf(p0)), fabsf(p1));      // (normalised units)
          S::template split<r>(p0, p1, out);
        });
        PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp + 512 * g), out.hi);
        PINN_STASH_ST(reinterpret_cast<u32x4*>(dsp + 512 * g + 256), out.lo);
      });
    } else {
      const float* hp = sx.act(0, H, lane);
      float* dsp = sx.dact(0, H, lane);
      f32x4 hl[NT];
#pragma unroll
      for (int t = 0; t < NT; ++t) load_block(hp, t, hl[t]);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          dh[t][r] = one(hl[t][r], dh[t][r]);
          if constexpr (S::kActScale != 1.0f) amax = fmaxf(amax, fabsf(dh[t][r]));      // (already in true units)
        }
        store_block(dsp, t, dh[t]);
      }
    }
  }
  if constexpr (S::kActScale != 1.0f) amax = fmaxf(amax, mx * unnorm);
}

}  // namespace x6
}  // namespace pinn
