// Does ONE wave overlap its own VALU instructions with its own bf16 MFMAs on gfx950?  (tools/mfma_valu_share.hip shows
// that two DIFFERENT waves of a SIMD do.)  One wave per SIMD; per iteration 16 v_mfma_f32_16x16x32_bf16 on 8 independent
// accumulators and NV integer VALU instructions, either grouped (16 MFMAs, then the VALU block) or interleaved
// (1 MFMA, NV/16 VALU, ...), pinned with inline asm.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int MODE>   // 0: MFMA only, 1: VALU only, 2: grouped, 3: interleaved
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  f32x4 acc[8];
  for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
  bf16x8 ab, bb;
  for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(1.0f + j * 0.01f + threadIdx.x * 1e-3f); bb[j] = (__bf16)(0.5f + j * 1e-3f); }
  unsigned v[8];
  for (int t = 0; t < 8; ++t) v[t] = threadIdx.x * 2654435761u + t;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (MODE != 1) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[t & 7]) : "v"(ab), "v"(bb));
      if (MODE == 3 || MODE == 1) {
#pragma unroll
        for (int q = 0; q < 3; ++q) asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(v[(3 * t + q) & 7]) : "v"(v[(3 * t + q + 1) & 7]));
      }
    }
    if (MODE == 2) {
#pragma unroll
      for (int t = 0; t < 48; ++t) asm volatile("v_xor_b32 %0, %0, %1\n\tv_add_u32 %0, %0, %1" : "+v"(v[t & 7]) : "v"(v[(t + 1) & 7]));
    }
  }
  float s = 0;
  for (int t = 0; t < 8; ++t) s += acc[t][0] + (float)v[t];
  out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int MODE>
static float run(float* out, int iters) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}
int main() {
  float* out; hipMalloc(&out, 256 * 256 * 4);
  const int iters = 40000;
  printf("per wave and iteration: 16 bf16 MFMAs (256 matrix-core cycles) + 96 VALU instructions (384 issue cycles)\n");
  printf("MFMA only %.3f ms | VALU only %.3f ms | grouped %.3f ms | interleaved %.3f ms\n", run<0>(out, iters), run<1>(out, iters), run<2>(out, iters), run<3>(out, iters));
  return 0;
}
