// Microbenchmark: what does a dense f32-MFMA stream sustain on this chip, alone and with the
// LDS A-fragment reads of the chain kernels?  (tools/, not part of the library)
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>   // 0: 16x16x4 regs only, 1: 16x16x4 + ds_read_b128 per 4 MFMA, 2: 32x32x2 regs only, 3: 32x32x2 + ds_read
__global__ __launch_bounds__(256) void k(float* out, int iters) {
  __shared__ __attribute__((aligned(16))) float lds[8192];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 0.001f * (i & 63);
  __syncthreads();
  if (MODE < 2) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float b = 0.5f + lane * 1e-3f;
    f32x4 a = {1.f, 0.5f, 0.25f, 0.125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 8; t += 2) {
        if (MODE == 1) a = *reinterpret_cast<const f32x4*>(&lds[((it * 8 + t) * 64 + lane) * 4 & 8191 & ~3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b, acc[t], 0, 0, 0);
          acc[t + 1] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[r], b, acc[t + 1], 0, 0, 0);
        }
      }
    }
    float s = 0;
    for (int t = 0; t < 8; ++t) s += acc[t][0] + acc[t][3];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  } else {
    f32x16 acc[4];
    for (int t = 0; t < 4; ++t)
      for (int r = 0; r < 16; ++r) acc[t][r] = 0;
    float b = 0.5f + lane * 1e-3f;
    f32x4 a = {1.f, 0.5f, 0.25f, 0.125f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int t = 0; t < 4; ++t) {
        if (MODE == 3) a = *reinterpret_cast<const f32x4*>(&lds[((it * 4 + t) * 64 + lane) * 4 & 8191 & ~3]);
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[r], b, acc[t], 0, 0, 0);
      }
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) s += acc[t][0] + acc[t][7];
    out[blockIdx.x * 256 + threadIdx.x] = s;
  }
}

template <int MODE>
void run(const char* name, int blocks_per_cu) {
  int cus = 256;
  float* out;
  hipMalloc(&out, (size_t)cus * blocks_per_cu * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // MFMAs per wave: 16x16x4: iters*32 (2048 FLOP each); 32x32x2: iters*16 (4096 FLOP each)
    double flop = (double)cus * blocks_per_cu * 4 * iters * (MODE < 2 ? 32.0 * 2048 : 16.0 * 4096);
    if (rep == 2) printf("%-28s blocks/CU %d: %.3f ms  %.1f TFLOP/s\n", name, blocks_per_cu, ms, flop / ms / 1e9);
  }
  hipFree(out);
}

int main() {
  for (int b = 1; b <= 2; ++b) {
    run<0>("16x16x4 regs", b);
    run<1>("16x16x4 + ds_read_b128/4", b);
    run<2>("32x32x2 regs", b);
    run<3>("32x32x2 + ds_read_b128/4", b);
  }
  return 0;
}
