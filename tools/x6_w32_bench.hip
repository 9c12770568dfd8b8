// Multiply phase of an x6 slab step restructured for ONE wave per SIMD with 32-row tiles and v_mfma_f32_32x32x16_bf16:
// per wave and slab 8 output tiles x (2 k-steps x 6 products) = 96 MFMAs of 32 cycles, 48 ds_read_b128, 12 LDS-DMA pieces,
// one barrier -- and F filler VALU instructions in every MFMA gap (the activation preparation of the real kernel would
// sit there).  Prints cycles per slab step (MFMA floor: 3072) for F = 0 .. 6, with / without the quarter-rate and
// transcendental share of the real preparation mix.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I physics-*/csrc -I include -o x6_w32_bench tools/x6_w32_bench.hip
#include "pinn_x6_core.h"
#include <cstdio>
using namespace pinn;
using namespace pinn::x6;

typedef float f32x16 __attribute__((ext_vector_type(16)));
#define MFMA32(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_bf16((a), (b), (c), 0, 0, 0)

struct A6 {
  bf16x8 f[6];   // [k-step][part]: 0..2 = s0 (h, m, l), 3..5 = s1
};
template <int T>
__device__ __forceinline__ void load_part(A6& a, int idx, unsigned addr0, unsigned addr1) {
  constexpr int kCopy = kSlabBytes / 3;
  switch (idx) {
    case 0: a.f[0] = lds_read_b128<T * 2048>(addr0); break;
    case 1: a.f[1] = lds_read_b128<kCopy + T * 2048>(addr0); break;
    case 2: a.f[2] = lds_read_b128<2 * kCopy + T * 2048>(addr0); break;
    case 3: a.f[3] = lds_read_b128<T * 2048>(addr1); break;
    case 4: a.f[4] = lds_read_b128<kCopy + T * 2048>(addr1); break;
    default: a.f[5] = lds_read_b128<2 * kCopy + T * 2048>(addr1); break;
  }
}
template <int N>
__device__ __forceinline__ void wait_all(A6& a) {
  asm volatile("s_waitcnt lgkmcnt(%6)" : "+v"(a.f[0]), "+v"(a.f[1]), "+v"(a.f[2]), "+v"(a.f[3]), "+v"(a.f[4]), "+v"(a.f[5]) : "n"(N));
}

struct Fill {
  float x[8];
  unsigned long long q;
  unsigned qa;
};
// F plain VALU + (MIX) one transcendental every third gap and one 32x32->64 multiply every fifth
template <int F, bool MIX, int GAP>
__device__ __forceinline__ void filler(Fill& f) {
#pragma unroll
  for (int j = 0; j < F; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f.x[(GAP * F + j) & 7]) : "v"(1.0001f), "v"(0.5f));
  if (MIX && F > 0) {
    if (GAP % 3 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f.x[GAP & 7]));
    if (GAP % 5 == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(f.q) : "v"(f.qa), "v"(0xD2511F53u) : "vcc");
  }
}

template <int F, bool MIX, bool DMA>
__global__ __launch_bounds__(256, 1) void k32(float* out, const __bf16* packed, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * kSlabBytes];
  for (int i = threadIdx.x; i < 2 * kSlabBytes / 2; i += blockDim.x) reinterpret_cast<__bf16*>(smem)[i] = (__bf16)(0.001f * (i % 977) - 0.4f);
  __syncthreads();
  Pipe6 pipe;
  pipe.lds = smem; pipe.par = 0;
  pipe.init(packed, 1 << 20, threadIdx.x);
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  bf16x8 bh[2], bm[2], bl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[s][j] = (__bf16)(0.1f + 0.01f * j + 1e-3f * lane); bm[s][j] = (__bf16)(1e-3f * j); bl[s][j] = (__bf16)(1e-5f * (j + s)); }
  Fill fl;
#pragma unroll
  for (int j = 0; j < 8; ++j) fl.x[j] = 0.01f * (lane + j);
  fl.q = lane; fl.qa = lane * 2654435761u;
  const Mat m{0u, 4};
  const int row = lane & 31, half = lane >> 5;
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    const char* slab = pipe.cur();
    const unsigned addr0 = (unsigned)(unsigned long long)(lptr_t)(slab + row * 64 + ((half ^ swz(row)) << 4));
    const unsigned addr1 = addr0 ^ 32u;
    A6 a[2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 6; ++i) load_part<0>(a[0], i, addr0, addr1);
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      A6& cur = a[t & 1];
      A6& nxt = a[(t + 1) & 1];
      if constexpr (t + 1 < 8) wait_all<0>(cur); else wait_all<0>(cur);
      static_for<12>([&](auto ic) {
        constexpr int i = decltype(ic)::value, s = i / 6, p = i % 6, gap = t * 12 + i;
        // products of a k-step in the order of mfma6: (l,h) (h,l) (m,m) (m,h) (h,m) (h,h)
        const bf16x8& av = p == 0 ? cur.f[3 * s + 2] : (p == 2 || p == 3) ? cur.f[3 * s + 1] : cur.f[3 * s];
        const bf16x8& bv = (p == 0 || p == 3 || p == 5) ? bh[s] : (p == 1 ? bl[s] : bm[s]);
        acc[t] = MFMA32(av, bv, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < 8 && (i & 1) == 0) load_part<t + 1>(nxt, i / 2, addr0, addr1);
        if constexpr (DMA && gap % 8 == 3) pipe.piece<8, 4>(m, it & 7, gap / 8, pipe.par ^ 1);
        filler<F, MIX, gap>(fl);
        __builtin_amdgcn_sched_barrier(0);
      });
    });
    if (DMA) { __syncthreads(); pipe.par ^= 1; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
#pragma unroll
  for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][7] + acc[t][15];
#pragma unroll
  for (int j = 0; j < 8; ++j) s += fl.x[j];
  s += (float)fl.q;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int F, bool MIX, bool DMA>
void run(float* out, __bf16* packed, unsigned long long* cyc) {
  const int iters = 2000;
  unsigned long long h = 0;
  float ms = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<F, MIX, DMA>), dim3(256), dim3(256), 0, 0, out, packed, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("F %d mix %d dma %d: %.0f ticks per slab step (floor 3072); wall %.3f ms = %.0f ns per step; clock %.2f GHz\n", F, (int)MIX, (int)DMA,
         (double)h / iters, ms, ms * 1e6 / iters, (double)h / (ms * 1e6));
}

int main() {
  float* out; __bf16* packed; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&packed, 8 << 20); hipMemset(packed, 0, 8 << 20); hipMalloc(&cyc, 8);
  run<0, false, false>(out, packed, cyc);
  run<0, false, true>(out, packed, cyc);
  run<1, false, true>(out, packed, cyc);
  run<2, false, true>(out, packed, cyc);
  run<3, false, true>(out, packed, cyc);
  run<4, false, true>(out, packed, cyc);
  run<5, false, true>(out, packed, cyc);
  run<2, true, true>(out, packed, cyc);
  run<3, true, true>(out, packed, cyc);
  run<4, true, true>(out, packed, cyc);
  return 0;
}
