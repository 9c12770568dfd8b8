// Cycles per slab of the x6 multiply phase (slab_mfma<16>: 96 bf16 MFMAs + 48 ds_read_b128) in isolation:
// 4 or 8 waves per CU (1 or 2 per SIMD), with / without LDS-DMA pieces in the loop.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I physics-*/csrc -I include -o x6_slab_bench tools/x6_slab_bench.hip
#include "pinn_x6_core.h"
#include <cstdio>
using namespace pinn;
using namespace pinn::x6;

template <bool DMA>
__global__ __launch_bounds__(512, 2) void k(float* out, const __bf16* packed, int iters, unsigned long long* cyc) {
  __shared__ __attribute__((aligned(1024))) char smem[2 * kSlabBytes];
  for (int i = threadIdx.x; i < 2 * kSlabBytes / 2; i += blockDim.x) reinterpret_cast<__bf16*>(smem)[i] = (__bf16)(0.001f * (i % 977) - 0.4f);
  __syncthreads();
  Pipe6 pipe;
  pipe.lds = smem; pipe.par = 0;
  pipe.init(packed, 1 << 20, threadIdx.x);
  const int lane = threadIdx.x & 63;
  f32x4 acc[16];
  for (int t = 0; t < 16; ++t) acc[t] = f32x4{0, 0, 0, 0};
  f32x4 v0 = {0.1f + lane * 1e-3f, 0.2f, -0.3f, 0.4f}, v1 = {0.5f, -0.6f, 0.7f, 0.8f + lane * 1e-3f};
  const Frag3 b = split3(v0, v1);
  const Mat m{0u, 4};
  unsigned long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it) {
    auto dma = [&](auto slotc) { constexpr int slot = decltype(slotc)::value; if (DMA && slot < 6) pipe.piece<8>(m, it & 7, slot, pipe.par ^ 1); };
    slab_mfma<X6, 16>(acc, b, pipe.cur(), lane, [](auto) {}, dma);
    if (DMA) { __syncthreads(); pipe.par ^= 1; }
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  float s = 0;
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

int main() {
  float* out; __bf16* packed; unsigned long long* cyc;
  hipMalloc(&out, 256 * 512 * 4); hipMalloc(&packed, 8 << 20); hipMemset(packed, 0, 8 << 20); hipMalloc(&cyc, 8);
  const int iters = 2000;
  for (int dma = 0; dma < 2; ++dma)
    for (int threads : {256, 512}) {
      unsigned long long h = 0;
      float ms = 0;
      hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
      for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        if (dma) hipLaunchKernelGGL(k<true>, dim3(256), dim3(threads), 0, 0, out, packed, iters, cyc);
        else hipLaunchKernelGGL(k<false>, dim3(256), dim3(threads), 0, 0, out, packed, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
        hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
      }
      printf("dma %d, %d waves/SIMD: wave 0 sees %.0f ticks per slab; wall %.3f ms = %.0f ns per slab step (MFMA-bound: %.0f ns at 2.4 GHz)", dma, threads / 256,
             (double)h / iters, ms, ms * 1e6 / iters, 1536.0 * threads / 256 / 2.4);
      // with a barrier per slab (dma = 1) or one wave per SIMD, wave 0 spans the whole kernel: ticks / wall = shader clock
      if (dma || threads == 256) printf("  -> shader clock %.2f GHz", (double)h / (ms * 1e6));
      printf("\n");
    }
  return 0;
}
