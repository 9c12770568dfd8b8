// Do f32 MFMA and VALU work share an execution resource on gfx950?  Half of the workgroups run a
// dense v_mfma_f32_16x16x4_f32 stream, the other half a dense VALU stream of one kind (2 workgroups
// per CU, one of each, so every SIMD hosts one MFMA wave and one VALU wave).  If the pipes were
// independent, t(both) ~= max(t_mfma, t_valu); if shared, ~= sum.
//   kinds: 0 v_fma_f32, 1 integer (v_mad_u64_u32 + xor, the Philox mix), 2 transcendental (v_exp_f32 / v_rcp_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KIND>
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
  } else if (KIND == 0) {
    float v[16];
    for (int t = 0; t < 16; ++t) v[t] = threadIdx.x * 1e-3f + t;
    float m = 0.999f, c = 1e-3f;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) v[t] = fmaf(v[t], m, c);
    }
    for (int t = 0; t < 16; ++t) s += v[t];
  } else if (KIND == 1) {
    unsigned v[8];
    for (int t = 0; t < 8; ++t) v[t] = threadIdx.x * 2654435761u + t;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        const unsigned long long p = (unsigned long long)0xD2511F53u * v[t];
        v[t] = (unsigned)(p >> 32) ^ (unsigned)p ^ 0x9E3779B9u;
      }
    }
    unsigned x = 0;
    for (int t = 0; t < 8; ++t) x ^= v[t];
    s = (float)x;
  } else {
    float v[8];
    for (int t = 0; t < 8; ++t) v[t] = 0.5f + threadIdx.x * 1e-3f + t * 0.01f;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[t]));
    }
    for (int t = 0; t < 8; ++t) s += v[t];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND>
static float run(int blocks, int im, int iv, int mask, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, im, iv, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int KIND>
static void test(const char* name, int iv, float* out) {
  const int im = 40000;
  float tm = run<KIND>(256, im, iv, 1, out), tv = run<KIND>(256, im, iv, 2, out), tb = run<KIND>(512, im, iv, 3, out);
  printf("%-28s MFMA-only %.3f ms | VALU-only %.3f ms | both %.3f ms | sum %.3f max %.3f\n", name, tm, tv, tb, tm + tv, tm > tv ? tm : tv);
}

int main() {
  float* out; hipMalloc(&out, 512 * 256 * 4);
  test<0>("f32 FMA", 80000, out);
  test<1>("int mad_u64_u32 + xor", 40000, out);
  test<2>("transcendental exp2 + rcp", 40000, out);
  return 0;
}
