// Do f32 MFMA and VALU work share an execution resource on gfx950?  Half of the workgroups run a
// dense v_mfma_f32_16x16x4_f32 stream, the other half a dense VALU stream of one kind (2 workgroups
// per CU, one of each, so every SIMD hosts one MFMA wave and one VALU wave).  If the pipes were
// independent, t(both) ~= max(t_mfma, t_valu); if shared, ~= sum.
//   kinds: 0 v_fma_f32, 1 integer (v_mad_u64_u32 + xor, the Philox mix), 2 transcendental (v_exp_f32 / v_rcp_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

template <int KIND, int MM>
__global__ __launch_bounds__(256) void k(float* out, int iters_m, int iters_v, int role_mask) {
  const int role = (role_mask == 3) ? (blockIdx.x >= gridDim.x / 2 ? 2 : 1) : role_mask;   // 1 = MFMA, 2 = VALU
  float s = 0;
  if (role == 1) {
    f32x4 acc[8];
    for (int t = 0; t < 8; ++t) acc[t] = f32x4{0, 0, 0, 0};
    float b = 0.5f + threadIdx.x * 1e-3f, a = 1.0f;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(1.0f + j * 0.01f); bb[j] = (__bf16)(b * 1e-2f + j * 1e-3f); }
    for (int it = 0; it < iters_m; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) {
        if (MM == 0) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        else if (MM == 1) { acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[t], 0, 0, 0); acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, acc[t], 0, 0, 0); }
        else if (MM == 2) { acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[t & 1], 0, 0, 0); acc[t & 1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, acc[t & 1], 0, 0, 0); }   // two chains, pairs back to back
        else { acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, acc[0], 0, 0, 0); acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bb, ab, acc[0], 0, 0, 0); }   // one dependent chain
      }
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
  } else if (KIND == 2) {
    float v[8];
    for (int t = 0; t < 8; ++t) v[t] = 0.5f + threadIdx.x * 1e-3f + t * 0.01f;
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 8; ++t) v[t] = __builtin_amdgcn_rcpf(__builtin_amdgcn_exp2f(v[t]));
    }
    for (int t = 0; t < 8; ++t) s += v[t];
  } else {
    // pinned single instructions (inline asm): 3 v_fma_f32, 4 v_pk_fma_f32, 5 v_add_f32, 6 v_mul_f32, 7 v_cvt_pk_bf16_f32,
    // 8 v_xor_b32, 9 v_cndmask_b32, 10 v_sub_f32 ... 16 independent registers, one instruction each per iteration
    float v[16];
    for (int t = 0; t < 16; ++t) v[t] = threadIdx.x * 1e-3f + t;
    float m = 0.999f, c = 1e-3f;
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 w[8], mm = {m, m}, cc = {c, c};
    for (int t = 0; t < 8; ++t) w[t] = f32x2{v[2 * t], v[2 * t + 1]};
    for (int it = 0; it < iters_v; ++it) {
#pragma unroll
      for (int t = 0; t < 16; ++t) {
        if (KIND == 3) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(v[t]) : "v"(m), "v"(c));
        if (KIND == 4 && t < 8) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(w[t]) : "v"(mm), "v"(cc));
        if (KIND == 5) asm volatile("v_add_f32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
        if (KIND == 6) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[t]) : "v"(m));
        if (KIND == 7) asm volatile("v_cvt_pk_bf16_f32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
        if (KIND == 8) asm volatile("v_xor_b32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
        if (KIND == 9) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(v[t]) : "v"(c) : "vcc");
        if (KIND == 10) asm volatile("v_lshlrev_b32 %0, 1, %0" : "+v"(v[t]));
        if (KIND == 11) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
        if (KIND == 12) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
        if (KIND == 13) asm volatile("v_and_b32 %0, %0, %1" : "+v"(v[t]) : "v"(c));
      }
    }
    for (int t = 0; t < 16; ++t) s += v[t];
    for (int t = 0; t < 8; ++t) s += w[t][0] + w[t][1];
  }
  out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int MM>
static float run(int blocks, int im, int iv, int mask, float* out) {
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 3; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<KIND, MM>), dim3(blocks), dim3(256), 0, 0, out, im, iv, mask);
    hipEventRecord(e1); hipEventSynchronize(e1);
    hipEventElapsedTime(&ms, e0, e1);
  }
  return ms;
}

template <int KIND, int MM>
static void test(const char* name, int iv, float* out) {
  const int im = 40000;
  float tm = run<KIND, MM>(256, im, iv, 1, out), tv = run<KIND, MM>(256, im, iv, 2, out), tb = run<KIND, MM>(512, im, iv, 3, out);
  printf("%-28s MFMA-only %.3f ms | VALU-only %.3f ms | both %.3f ms | sum %.3f max %.3f\n", name, tm, tv, tb, tm + tv, tm > tv ? tm : tv);
}

int main() {
  float* out; hipMalloc(&out, 512 * 256 * 4);
  printf("-- v_mfma_f32_16x16x4_f32 stream\n");
  test<0, 0>("f32 FMA", 80000, out);
  test<1, 0>("int mad_u64_u32 + xor", 40000, out);
  test<2, 0>("transcendental exp2 + rcp", 40000, out);
  printf("-- v_mfma_f32_16x16x32_bf16 stream (2 per f32 one: same MFMA-only time)\n");
  test<0, 1>("f32 FMA", 80000, out);
  test<1, 1>("int mad_u64_u32 + xor", 40000, out);
  test<2, 1>("transcendental exp2 + rcp", 40000, out);
  printf("-- same, MFMAs in two dependent chains (pairs back to back)\n");
  test<1, 2>("int mad_u64_u32 + xor", 40000, out);
  test<2, 2>("transcendental exp2 + rcp", 40000, out);
  test<3, 2>("v_fma_f32", 40000, out);
  printf("-- same, MFMAs in ONE dependent chain\n");
  test<1, 3>("int mad_u64_u32 + xor", 40000, out);
  test<2, 3>("transcendental exp2 + rcp", 40000, out);
  test<3, 3>("v_fma_f32", 40000, out);
  printf("-- single pinned instructions against the independent bf16 stream\n");
  test<3, 1>("v_fma_f32", 40000, out);
  test<4, 1>("v_pk_fma_f32", 40000, out);
  test<5, 1>("v_add_f32", 40000, out);
  test<6, 1>("v_mul_f32", 40000, out);
  test<7, 1>("v_cvt_pk_bf16_f32", 40000, out);
  test<8, 1>("v_xor_b32", 40000, out);
  test<9, 1>("v_cndmask_b32", 40000, out);
  test<10, 1>("v_lshlrev_b32", 40000, out);
  test<11, 1>("v_mul_lo_u32", 10000, out);
  test<12, 1>("v_sub_f32", 40000, out);
  test<13, 1>("v_and_b32", 40000, out);
  return 0;
}
