// pinn_x6.hip -- fp32-accurate forward / MC-dropout kernels on the bf16 matrix cores
// (pinn_net_t.precision = PINN_PREC_F32X6).  See pinn_x6_core.h.
#include <cstdlib>
#include "pinn_x6_core.h"

namespace pinn {
namespace x6 {

#ifdef PINN_CLOCK_STAMP
// diagnostic build only (tools/clock_in_kernel.py; MI355X_MICROARCH.md "DVFS give-back" item 6): s_memtime (shader cycles) and
// s_memrealtime (100 MHz) of every workgroup at the start and the end of mlp_x6_kernel -> in-kernel clock.  The values go to a
// buffer nothing else reads.
__device__ unsigned long long g_clock_stamps[1024][4];
__device__ __forceinline__ void clock_stamp(int slot) {
  unsigned long long c, r;
  asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
  if (threadIdx.x == 0 && blockIdx.x < 1024) { g_clock_stamps[blockIdx.x][slot] = c; g_clock_stamps[blockIdx.x][slot + 1] = r; }
}
#endif

struct PackJobs6 {
  PackJob j[18];
  int n;
  long long copy_stride;
  int with_f16;          // also the two fp16 copies (scheme X3)
  int with_bf16;         // the three bf16 copies (scheme X6: only PINN_PREC_F32X6_G6's gradient kernels read them)
};

static int cu_count_x() { return cu_count_cached(); }

// three bf16 copies (hi, mid, lo with w = hi + mid + lo exactly) of every matrix and its transpose,
// K permuted inside each 32-group (pack_col_x6)
// Range record (pinn_net_range_status): scheme X3's fp16 copies hold 64 w, finite only while |w| < 1023.5; every workgroup
// writes whether it met a weight outside that (or a non-finite one) into its own word of `status` -- plain stores, rewritten
// by every call -- and workgroup (0, 0) clears the word of the gradient check that a training call's finalize kernel sets.
__global__ __launch_bounds__(256) void pack_x6_kernel(const float* __restrict__ params, __bf16* __restrict__ packed, PackJobs6 jobs,
                                                      unsigned* __restrict__ status, unsigned* __restrict__ zero_word) {
  if (zero_word && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) *zero_word = 0u;      // TrainBuffers::emax, for the forward kernel behind us
  const PackJob j = jobs.j[blockIdx.y];
  const long long n2 = (long long)j.rows * j.Kp / 2;      // pairs of consecutive packed positions (Kp is a multiple of 64)
  int bad = 0;
  auto src_of = [&](int row, int q) -> float {
    const int k = (q & ~31) + pack_col_x6(q & 31);
    if (k >= j.K) return 0.0f;
    return j.transposed ? params[j.src + (long long)k * j.src_ld + row] : params[j.src + (long long)row * j.src_ld + k];
  };
  for (long long e2 = (long long)blockIdx.x * blockDim.x + threadIdx.x; e2 < n2; e2 += (long long)gridDim.x * blockDim.x) {
    const long long e = 2 * e2;
    const int row = (int)(e / j.Kp), q = (int)(e % j.Kp);
    const float v[2] = {src_of(row, q), src_of(row, q + 1)};
    // group-major: [K-group q / 32][row][32] (pinn_x6_core.h PipeT::piece)
    const long long at = (long long)(q >> 5) * j.rows * 32 + (long long)row * 32 + (q & 31);
    if (jobs.with_bf16) {          // three bf16 copies, w = hi + mid + lo exactly (scheme X6: PINN_PREC_F32X6_G6's gradients)
      __bf16 h[2], m[2], l[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        h[t] = (__bf16)v[t];
        const float r1 = v[t] - (float)h[t];
        m[t] = (__bf16)r1;
        l[t] = (__bf16)(r1 - (float)m[t]);
      }
      typedef __bf16 bf16x2_t __attribute__((ext_vector_type(2)));
      *reinterpret_cast<bf16x2_t*>(packed + j.dst + at) = bf16x2_t{h[0], h[1]};
      *reinterpret_cast<bf16x2_t*>(packed + jobs.copy_stride + j.dst + at) = bf16x2_t{m[0], m[1]};
      *reinterpret_cast<bf16x2_t*>(packed + 2 * jobs.copy_stride + j.dst + at) = bf16x2_t{l[0], l[1]};
    }
    // every matrix again as two fp16 copies of X3::kWScale * w (scheme X3: forward and backward chain), behind the three
    // bf16 copies: same element offsets, same K permutation
    if (jobs.with_f16) {
      _Float16* p16 = reinterpret_cast<_Float16*>(packed + 3 * jobs.copy_stride);
      _Float16 h16[2], l16[2];
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        const float vs = v[t] * X3::kWScale;
        bad |= !(fabsf(vs) <= 65504.0f);           // (also true for a NaN)
        h16[t] = (_Float16)vs;
        l16[t] = (_Float16)(vs - (float)h16[t]);
      }
      typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));
      *reinterpret_cast<f16x2_t*>(p16 + j.dst + at) = f16x2_t{h16[0], h16[1]};
      *reinterpret_cast<f16x2_t*>(p16 + jobs.copy_stride + j.dst + at) = f16x2_t{l16[0], l16[1]};
    }
  }
  if (status) {
    bad = __syncthreads_or(bad);
    if (threadIdx.x == 0) {
      status[blockIdx.y * gridDim.x + blockIdx.x] = bad ? 1u : 0u;
      if (blockIdx.x == 0 && blockIdx.y == 0) status[gridDim.x * gridDim.y] = 0u;
    }
  }
}

void launch_pack_x6(const pinn_net_t* net, const float* d_params, hipStream_t st, unsigned* zero_word) {
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  PackLayout K{H, nh};
  PackJobs6 jobs;
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
  jobs.copy_stride = K.total();
  jobs.with_f16 = 1;
  jobs.with_bf16 = net->precision != PINN_PREC_F32X6;      // (_G6: scheme X6 gradients; PINN_PREC_BF16 on the wide nets: the first copy)
  static_assert(kRangeStatusBytes >= (18 * kRangePackBlocks + 1) * sizeof(unsigned), "range record too small");
  hipLaunchKernelGGL(pack_x6_kernel, dim3(kRangePackBlocks, n), dim3(256), 0, st, d_params, (__bf16*)net->d_packed, jobs, range_status_words(net), zero_word);
}

// WAVES = 8: 128-row tiles, two waves per SIMD.  WAVES = 4 (small row counts, fewer tiles than CUs): 64-row tiles, one
// wave per SIMD -- twice the CUs busy and no wave shares its SIMD's matrix core, so a tile finishes in about half the time.
// MC-dropout moments: the passes are accumulated in TWO Welford states by parity (passes 0, 2, 4 .. and 1, 3, 5 ..) and merged
// at the end (Chan's formula, one fixed order) -- in every variant, so that the results do not depend on which one ran.
// kSplit (MC, 8 waves, row counts with at most one 64-row tile per CU -- the reference's own sizes): the two parities are two
// WAVES.  A lone wave per SIMD issues a pass's MFMAs and its activation arithmetic in order (24.7 us per pass at 1.1e4 rows,
// 46 % of the large-batch rate); passes are independent, so waves 4-7 run the odd passes of the rows whose even passes waves
// 0-3 run, in step on the same weight stream, and the second wave of a SIMD fills the first one's gaps.
struct Moments {
  float mean, m2, sl;
};
__device__ __forceinline__ void merge_moments(Moments& a, const Moments& b, int n_passes) {      // a: even passes, b: odd passes
  const float n0 = (float)((n_passes + 1) / 2), n1 = (float)(n_passes / 2), inv_n = 1.0f / (float)n_passes;
  const float d = b.mean - a.mean;
  a.mean = fmaf(d, n1 * inv_n, a.mean);
  a.m2 = (a.m2 + b.m2) + (d * d) * (n0 * n1 * inv_n);
  a.sl = a.sl + b.sl;
}
template <int H, bool MC, bool kBits, int WAVES, bool kSplit = false>
__global__ __launch_bounds__(WAVES * 64, WAVES / 4) void mlp_x6_kernel(FwdArgs a, const __bf16* packed) {
  using S = X3;                                   // forward passes: two fp16 parts, three products (pinn_x6_core.h)
  static_assert(!kSplit || (MC && WAVES == 8), "the pass split is an MC-dropout variant of the 8-wave kernel");
  constexpr int kThreadsX = WAVES * 64, kTileRowsX = kSplit ? 64 : WAVES * 16;
  // one LDS block, small things FIRST: a ds instruction's immediate offset is 16 bits, and every per-layer bias /
  // head-weight address beyond 64 KB would need its own address register (hipcc hoists them all: spills)
  constexpr int kSmallBytes = kMaxSmall * 4, kW0Bytes = 8 * kW0Stride * 4;
  constexpr int kMergeBytes = kSplit ? 4 * 16 * 3 * 4 : 0;      // the odd passes' moments of the tile's 64 rows
  constexpr int kSlabAt = (kSmallBytes + kW0Bytes + kMergeBytes + 1023) & ~1023;
  __shared__ __attribute__((aligned(1024))) char smem[kSlabAt + 2 * S::Pipe::kSlab];
  float* small = reinterpret_cast<float*>(smem);
  float* w0t = reinterpret_cast<float*>(smem + kSmallBytes);
  float* merge = reinterpret_cast<float*>(smem + kSmallBytes + kW0Bytes);
  char* lds_w = smem + kSlabAt;
  ParamLayout L{a.H, a.nh};
  PackLayout K{a.H, a.nh};
  fill_small<S, kThreadsX>(small, w0t, a.params, L);
  S::Pipe pipe;
  pipe.lds = lds_w;
  pipe.init(packed + 3 * K.total(), (unsigned)(K.total() * 2), threadIdx.x);      // the fp16 copies sit behind the three bf16 ones
  pipe.template prime<clog2(H), WAVES>(first_mat<H>(K));
#ifdef PINN_X6_STAMP
  for (int k = 0; k < 8; ++k) pipe.seg[k] = 0;
  pipe.last = stamp();
#endif

  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long long n_tiles = (a.n_rows + kTileRowsX - 1) / kTileRowsX;
#ifdef PINN_CLOCK_STAMP
  clock_stamp(0);
#endif
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const int row_wave = kSplit ? (wave & 3) : wave, parity = kSplit ? (wave >> 2) : 0;
    const long long lrow = tile * kTileRowsX + row_wave * kWaveRows + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const long long srow = valid ? lrow : a.n_rows - 1;
    const long long grow = a.drop.row_offset + lrow;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2];
    const f32x4 xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    RowCtx c{lane, lane >> 4, grow, srow, a.n_rows, 0u, a.drop.mode};
    if (!MC) {
      float u, z;
      PINN_STAMP(pipe, 3);
      forward_pass<S, H, kBits, false, WAVES>(w0t, small, L, pipe, a.drop, c, xa, xb, u, z);
      if (valid && lane < 16) {
        a.o0[lrow] = u;
        a.o1[lrow] = logf(softplus_f32(z) + 1e-6f);
      }
    } else {
      float u_eval = 0.f;
      Moments m_even{0.f, 0.f, 0.f}, m_odd{0.f, 0.f, 0.f};      // (kSplit: m_even = this wave's parity, m_odd = the other half's, handed over)
      // kSplit: this wave runs the eval pass and the passes of its parity -- ceil(T / 2) of them in BOTH halves of the workgroup (the
      // slab barriers count every wave): the odd half's last one is a spare when T is odd, computed and dropped
      const int t_step = kSplit ? 2 : 1, t_end = kSplit ? 2 * ((a.n_passes + 1) / 2) : a.n_passes;
#pragma unroll 1
      for (int t = -1; t < t_end; t = t < 0 ? parity : t + t_step) {
        const bool spare = t >= a.n_passes;
        c.mode = (t < 0) ? PINN_DROP_NONE : a.drop.mode;
        c.pass = (unsigned)(t < 0 || spare ? 0 : t);
        float u, z;
        PINN_STAMP(pipe, 3);
        forward_pass<S, H, kBits, false, WAVES>(w0t, small, L, pipe, a.drop, c, xa, xb, u, z);
        if (t < 0) {
          u_eval = u;
        } else if (!spare) {
          const float inv_k = 1.0f / (float)(t / 2 + 1), lv = logf(softplus_f32(z) + 1e-6f);      // k-th pass of its parity
          if (kSplit || !(t & 1)) { welford_update(m_even.mean, m_even.m2, u - u_eval, inv_k); m_even.sl += lv; }
          else { welford_update(m_odd.mean, m_odd.m2, u - u_eval, inv_k); m_odd.sl += lv; }
        }
      }
      if constexpr (kSplit) {        // the odd half hands its moments over
        float* slot = merge + (row_wave * 16 + (lane & 15)) * 3;
        if (parity == 1 && lane < 16) { slot[0] = m_even.mean; slot[1] = m_even.m2; slot[2] = m_even.sl; }
        __syncthreads();
        if (parity == 0) { m_odd.mean = slot[0]; m_odd.m2 = slot[1]; m_odd.sl = slot[2]; }
        __syncthreads();             // (the next tile's hand-over must not overtake this read)
      }
      if (valid && lane < 16 && parity == 0) {
        merge_moments(m_even, m_odd, a.n_passes);
        const float inv_t = 1.0f / (float)a.n_passes;
        const float var = m_even.m2 * inv_t;                     // population variance of the passes
        a.o0[lrow] = u_eval;
        a.o1[lrow] = expf(0.5f * (m_even.sl * inv_t));
        a.o2[lrow] = sqrtf(var);
      }
    }
  }
#ifdef PINN_X6_STAMP
  if (blockIdx.x == 0 && lane == 0)
    for (int k = 0; k < 8; ++k) g_x6_stamps[wave * 8 + k] = pipe.seg[k];
#endif
#ifdef PINN_CLOCK_STAMP
  clock_stamp(2);
#endif
}

}  // namespace x6

int launch_forward_x6(const pinn_net_t* net, const FwdArgs& a, bool mc, void* stream) {
  using namespace x6;
  hipStream_t st = (hipStream_t)stream;
  (void)hipGetLastError();
  launch_pack_x6(net, a.params, st, nullptr);
  const int cus = cu_count_x();
  const long long t128 = (a.n_rows + 127) / 128;
#ifdef PINN_DEBUG_HOOKS
  static const bool force8 = getenv("PINN_X6_WAVES8") != nullptr;     // measurement builds only: always the 8-wave kernels
#else
  constexpr bool force8 = false;
#endif
  const bool small_n = !force8 && 2 * t128 <= cus;           // 64-row tiles still fit one per CU
  const long long n_tiles = small_n ? (a.n_rows + 63) / 64 : t128;
  const int grid = (int)(n_tiles < cus ? n_tiles : cus);
  const __bf16* packed = (const __bf16*)net->d_packed;
  const bool bits = a.drop.mode == PINN_DROP_BITS;
#define PINN_LAUNCH_X(HH, MCC, BB)                                                                                              \
  do {                                                                                                                          \
    if (small_n && MCC) hipLaunchKernelGGL((mlp_x6_kernel<HH, MCC, BB, 8, MCC>), dim3(grid), dim3(512), 0, st, a, packed);      \
    else if (small_n) hipLaunchKernelGGL((mlp_x6_kernel<HH, MCC, BB, 4>), dim3(grid), dim3(256), 0, st, a, packed);             \
    else hipLaunchKernelGGL((mlp_x6_kernel<HH, MCC, BB, 8>), dim3(grid), dim3(512), 0, st, a, packed);                          \
  } while (0)
  if (net->hidden == 256) {
    if (mc) { if (bits) PINN_LAUNCH_X(256, true, true); else PINN_LAUNCH_X(256, true, false); }
    else    { if (bits) PINN_LAUNCH_X(256, false, true); else PINN_LAUNCH_X(256, false, false); }
  } else {
    if (mc) { if (bits) PINN_LAUNCH_X(128, true, true); else PINN_LAUNCH_X(128, true, false); }
    else    { if (bits) PINN_LAUNCH_X(128, false, true); else PINN_LAUNCH_X(128, false, false); }
  }
#undef PINN_LAUNCH_X
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

}  // namespace pinn

#ifdef PINN_CLOCK_STAMP
extern "C" int pinn_clock_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pinn::x6::g_clock_stamps), sizeof(unsigned long long) * 1024 * 4);
}
#endif
#ifdef PINN_X6_STAMP
extern "C" int pinn_x6_debug_read(unsigned long long* host) {
  return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(pinn::x6::g_x6_stamps), sizeof(unsigned long long) * 64);
}
#endif
