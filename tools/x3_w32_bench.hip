// Multiply phase of a scheme-X3 slab step (two fp16 parts, three products) restructured for ONE wave per SIMD with 32-row tiles
// and v_mfma_f32_32x32x16_f16: per wave and slab 8 output tiles x (2 k-steps x 3 products) = 48 MFMAs of 32 cycles (floor 1536),
// 32 ds_read_b128 (half the LDS bytes of two waves x 16 rows), 8 LDS-DMA pieces, one barrier -- and F filler VALU
// instructions in every MFMA gap, where the activation preparation of the real kernel would sit (the X3 forward kernels
// issue ~360 VALU instructions per SIMD and hidden-layer step = 7.5 per gap; their step takes 2820 cycles = 1.25 us at
// 2.26 GHz, tools/x6_stamps.py).  Prints cycles and ns per slab step for F = 0 .. 9 with the real mix's share of
// transcendental and quarter-rate instructions.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I physics-*/csrc -I include -o x3_w32_bench tools/x3_w32_bench.hip
#include "pinn_x6_core.h"
#include <cstdio>
using namespace pinn;
using namespace pinn::x6;

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 h16x8 __attribute__((ext_vector_type(8)));
#define MFMA32H(a, b, c) __builtin_amdgcn_mfma_f32_32x32x16_f16((a), (b), (c), 0, 0, 0)
using Pipe2 = PipeT<2>;

struct A4 {
  h16x8 f[4];   // [k-step][part]: 0, 1 = s0 (h, l), 2, 3 = s1
};
template <int OFF>
__device__ __forceinline__ h16x8 rd(unsigned addr) {
  u32x4 v;
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(v) : "v"(addr), "n"(OFF));
  return __builtin_bit_cast(h16x8, v);
}
template <int T>
__device__ __forceinline__ void load_part(A4& a, int idx, unsigned addr0, unsigned addr1) {
  switch (idx) {
    case 0: a.f[0] = rd<T * 2048>(addr0); break;
    case 1: a.f[1] = rd<kCopyLds + T * 2048>(addr0); break;
    case 2: a.f[2] = rd<T * 2048>(addr1); break;
    default: a.f[3] = rd<kCopyLds + T * 2048>(addr1); break;
  }
}
__device__ __forceinline__ void wait_all(A4& a) {
  asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(a.f[0]), "+v"(a.f[1]), "+v"(a.f[2]), "+v"(a.f[3]));
}

struct Fill {
  float x[8];
  unsigned long long q;
  unsigned qa;
};
template <int F, bool MIX, int GAP>
__device__ __forceinline__ void filler(Fill& f) {
#pragma unroll
  for (int j = 0; j < F; ++j) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f.x[(GAP * F + j) & 7]) : "v"(1.0001f), "v"(0.5f));
  if (MIX && F > 0) {
    if (GAP % 2 == 0) asm volatile("v_exp_f32 %0, %0" : "+v"(f.x[GAP & 7]));       // ~1 transcendental per 7 instructions
    if (GAP % 3 == 0) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(f.q) : "v"(f.qa), "v"(0xD2511F53u) : "vcc");
  }
}

template <int F, bool MIX, bool DMA>
__global__ __launch_bounds__(256, 1) void k32(float* out, const __bf16* packed, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * Pipe2::kSlab];
  for (int i = threadIdx.x; i < 2 * Pipe2::kSlab / 2; i += blockDim.x) reinterpret_cast<_Float16*>(smem)[i] = (_Float16)(0.001f * (i % 977) - 0.4f);
  __syncthreads();
  Pipe2 pipe;
  pipe.lds = smem; pipe.par = 0;
  pipe.init(packed, 1 << 20, threadIdx.x);
  const int lane = threadIdx.x & 63;
  f32x16 acc[8];
#pragma unroll
  for (int t = 0; t < 8; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
  h16x8 bh[2], bl[2];
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int j = 0; j < 8; ++j) { bh[s][j] = (_Float16)(0.1f + 0.01f * j + 1e-3f * lane); bl[s][j] = (_Float16)(1e-4f * (j + s)); }
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
    A4 a[2];
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i) load_part<0>(a[0], i, addr0, addr1);
    __builtin_amdgcn_sched_barrier(0);
    static_for<8>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      A4& cur = a[t & 1];
      A4& nxt = a[(t + 1) & 1];
      wait_all(cur);
      static_for<6>([&](auto ic) {
        constexpr int i = decltype(ic)::value, s = i / 3, p = i % 3, gap = t * 6 + i;
        // products of a k-step: (l, h) (h, l) (h, h)
        const h16x8& av = p == 0 ? cur.f[2 * s + 1] : cur.f[2 * s];
        const h16x8& bv = p == 1 ? bl[s] : bh[s];
        acc[t] = MFMA32H(av, bv, acc[t]);
        __builtin_amdgcn_sched_barrier(0);
        if constexpr (t + 1 < 8 && i < 4) load_part<t + 1>(nxt, i, addr0, addr1);
        if constexpr (DMA && gap % 6 == 2) pipe.piece<8, 4>(m, it & 7, gap / 6, pipe.par ^ 1);
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
  const int iters = 4000;
  unsigned long long h = 0;
  float ms = 0;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 2; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k32<F, MIX, DMA>), dim3(256), dim3(256), 0, 0, out, packed, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
  }
  printf("F %d mix %d dma %d: %.0f ticks per slab step (floor 1536); wall %.3f ms = %.0f ns per step\n", F, (int)MIX, (int)DMA,
         (double)h / iters, ms, ms * 1e6 / iters);
}

int main() {
  float* out; __bf16* packed; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&packed, 8 << 20); hipMemset(packed, 0, 8 << 20); hipMalloc(&cyc, 8);
  run<0, false, false>(out, packed, cyc);
  run<0, false, true>(out, packed, cyc);
  run<3, true, true>(out, packed, cyc);
  run<5, true, true>(out, packed, cyc);
  run<6, true, true>(out, packed, cyc);
  run<7, true, true>(out, packed, cyc);
  run<8, true, true>(out, packed, cyc);
  run<9, true, true>(out, packed, cyc);
  return 0;
}
