// Do f32 MFMA and f32 VALU FMA share an execution resource on gfx950?  Half of the workgroups run a
// dense v_mfma_f32_16x16x4_f32 stream, the other half a dense v_fma_f32 stream (2 workgroups per CU,
// one of each).  If the pipes were independent, t(both) ~= max(t_mfma, t_valu); if shared, ~= sum.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void k(float* out, int iters_m, int iters_v, int role_mask) {
  const int role = (role_mask == 3) ? (blockIdx.x >= gridDim.x / 2 ? 2 : 1) : role_mask;   // 1 = MFMA, 2 = VALU
  float s = 0;
  if (role == 1) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float b = 0.5f + threadIdx.x * 1e-3f, a = 1.0f;
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    for (int t = 0; t < 8; ++t) s += acc[t][0];
  } else {
    float v[16];
    for (int t = 0; t < 16; ++t) v[t] = threadIdx.x * 1e-3f + t;
    float m = 0.999f, c = 1e-3f;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = fmaf(v[t], m, c);
    }
    for (int t = 0; t < 16; ++t) s += v[t];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

static float run(int blocks, int im, int iv, int mask, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, im, iv, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

int main() {
  float* out; hipMalloc(&out, 512 * 256 * 4);
  const int im = 40000, iv = 80000;     // 8 MFMA x 32 cyc = 256 cyc/iter ; 16 FMA x 4 cyc = 64 cyc/iter (one wave per SIMD)
  float tm = run(256, im, iv, 1, out), tv = run(256, im, iv, 2, out), tb = run(512, im, iv, 3, out);
  printf("MFMA-only %.3f ms | VALU-only %.3f ms | both (1 WG each per CU) %.3f ms | sum %.3f max %.3f\n", tm, tv, tb, tm + tv, tm > tv ? tm : tv);
  return 0;
}
