// How much of a VALU chunk hides under the other wave's MFMA group?  Two waves per SIMD (one 512-thread workgroup per CU),
// each alternating G bf16 MFMAs (one dependent chain) with K VALU ops -- the x6 chain's shape -- with / without a barrier
// every 16 groups, with AGPR accumulators, two interleaved chains, or the second wave of a SIMD running its chunk BEFORE its
// MFMAs.  Measured law (-DVMODE=0..5 = shift-add / xor-add / Philox-like multiply / exp+rcp / independent xor / dependent
// xor-add streams): t(both) ~= t(MFMA) + t(VALU) - c * min(...), c = 0.25-0.6, whatever the granularity, chain structure,
// accumulator file or barrier; the staggered variant is worse.  (The "roles" variant -- waves 0-3 only MFMAs, 4-7 only VALU,
// one of each per SIMD since wave w runs on SIMD w % 4 -- measures slower than the SUM of its parts and is not understood;
// tools/mfma_valu_share.hip, an MFMA-only and a VALU-only WORKGROUP per CU, overlaps ~100 %.)
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DVMODE=1 -o mfma_valu_alternate tools/mfma_valu_alternate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#ifndef ROLE_ODD
#define ROLE_ODD 0
#endif
#ifndef VMODE
#define VMODE 0
#endif
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int K, bool BAR, bool STAGGER, int G = 6, bool AG = false, bool IL = false, bool ROLE = false>
__global__ __launch_bounds__(512, 1) void k(float* out, int iters) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  bf16x8 a, b;
  for (int i = 0; i < 8; ++i) { a[i] = (__bf16)(0.01f * (lane + i)); b[i] = (__bf16)(0.02f * (lane - i)); }
  f32x4 acc[4] = {};
  unsigned v0 = lane * 2654435761u, v1 = lane + 17, v2 = lane ^ 0x55, v3 = 7 * lane;
  auto valu = [&]() {
#pragma unroll
    for (int q = 0; q < K / 4; ++q) {
#if VMODE == 0
      v0 = v0 * 3u + v1; v1 = (v1 ^ v2) + 0x9E3779B9u; v2 = v2 * 5u + v3; v3 = (v3 ^ v0) + 0x7F4A7C15u;   // shift-add / xor / add
#elif VMODE == 1
      v0 = (v0 ^ v1) + v2; v1 = (v1 ^ v2) + 0x9E3779B9u; v2 = (v2 >> 3) ^ v3; v3 = (v3 ^ v0) + 0x7F4A7C15u;   // no multiplies at all
#elif VMODE == 2
      v0 = (unsigned)(((unsigned long long)v0 * 0xD2511F53u) >> 32) ^ v1; v1 = v1 + 0x9E3779B9u;               // Philox-like: 32x32->64 multiply
      v2 = (unsigned)(((unsigned long long)v2 * 0xCD9E8D57u) >> 32) ^ v3; v3 = v3 + 0x7F4A7C15u;
#elif VMODE == 4
      asm volatile("v_xor_b32 %0, %0, %4\n\tv_xor_b32 %1, %1, %4\n\tv_xor_b32 %2, %2, %4\n\tv_xor_b32 %3, %3, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(lane));  // 4 independent ops
#elif VMODE == 5
      asm volatile("v_xor_b32 %0, %0, %4\n\tv_add_u32 %0, %0, %4\n\tv_xor_b32 %0, %0, %4\n\tv_add_u32 %0, %0, %4" : "+v"(v0), "+v"(v1), "+v"(v2), "+v"(v3) : "v"(lane));  // one dependent chain
#else
      { float f0 = __builtin_bit_cast(float, v0 | 0x3f000000u), f1 = __builtin_bit_cast(float, v1 | 0x3f000000u);
        f0 = __builtin_amdgcn_exp2f(f0); f1 = __builtin_amdgcn_rcpf(f1 + 1.0f); v0 = __builtin_bit_cast(unsigned, f0) ^ v2; v1 = __builtin_bit_cast(unsigned, f1) + v3; }
#endif
    }
  };
  auto mfma = [&](int t) {
#pragma unroll
    for (int q = 0; q < G; ++q) {
      if (IL) acc[(t & 1) * 2 + (q & 1)] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[(t & 1) * 2 + (q & 1)], 0, 0, 0);     // two chains interleaved
      else if (AG) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[t & 3]) : "v"(a), "v"(b));
      else acc[t & 3] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, acc[t & 3], 0, 0, 0);
    }
  };
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int t = 0; t < 16; ++t) {
      if (ROLE) { if (ROLE_ODD ? (wave & 1) : (wave >= 4)) { valu(); __builtin_amdgcn_sched_barrier(0); valu(); } else { mfma(t); __builtin_amdgcn_sched_barrier(0); mfma(t + 1); } }
      else if (STAGGER && wave >= 4) { valu(); __builtin_amdgcn_sched_barrier(0); mfma(t); }
      else { mfma(t); __builtin_amdgcn_sched_barrier(0); valu(); }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (BAR) __syncthreads();
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + (float)(v0 ^ v1 ^ v2 ^ v3);
}
template <int K, bool BAR, bool ST, int G = 6, bool AG = false, bool IL = false, bool ROLE = false>
void run(float* out, const char* name) {
  const int iters = 2000;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float ms = 0;
  for (int rep = 0; rep < 2; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k<K, BAR, ST, G, AG, IL, ROLE>), dim3(256), dim3(512), 0, 0, out, iters);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  }
  printf("%-28s K=%3d VALU per 6 MFMA: %7.0f ns per 16-group step  (MFMA only: 2 waves x 96 MFMA x 16 cyc = 1280 ns at 2.4 GHz)\n", name, K, ms * 1e6 / iters);
}
int main() {
  printf("VMODE %d\n", VMODE);
  float* out; (void)hipMalloc(&out, 256 * 512 * 4);
  run<16, false, false, 0>(out, "VALU only"); run<24, false, false, 0>(out, "VALU only"); run<32, false, false, 0>(out, "VALU only");
  run<0, false, false>(out, "no barrier");
  run<16, false, false>(out, "no barrier"); run<24, false, false>(out, "no barrier"); run<32, false, false>(out, "no barrier");
  run<16, true, false>(out, "barrier / 16 groups"); run<24, true, false>(out, "barrier / 16 groups"); run<32, true, false>(out, "barrier / 16 groups");
  run<0, false, false, 6, true>(out, "AGPR acc, no barrier"); run<16, false, false, 6, true>(out, "AGPR acc, no barrier");
  run<24, false, false, 6, true>(out, "AGPR acc, no barrier"); run<32, false, false, 6, true>(out, "AGPR acc, no barrier");
  run<0, false, false, 6, false, true>(out, "2 chains interleaved"); run<16, false, false, 6, false, true>(out, "2 chains interleaved");
  run<24, false, false, 6, false, true>(out, "2 chains interleaved"); run<32, false, false, 6, false, true>(out, "2 chains interleaved");
  run<32, false, false, 12, false, true>(out, "2 chains, 12 MFMA : K"); run<48, false, false, 12, false, true>(out, "2 chains, 12 MFMA : K"); run<64, false, false, 12, false, true>(out, "2 chains, 12 MFMA : K");
  run<32, false, false, 12, false, false>(out, "1 chain, 12 MFMA : K"); run<48, false, false, 12, false, false>(out, "1 chain, 12 MFMA : K"); run<64, false, false, 12, false, false>(out, "1 chain, 12 MFMA : K");
  run<8, false, false, 0>(out, "VALU only"); run<8, false, false>(out, "no barrier"); run<8, false, false, 6, false, false, true>(out, "roles: 4 MFMA + 4 VALU waves");
  run<12, false, false, 0>(out, "VALU only"); run<12, false, false>(out, "no barrier"); run<12, false, false, 6, false, false, true>(out, "roles: 4 MFMA + 4 VALU waves");
  run<16, false, false, 6, false, false, true>(out, "roles: 4 MFMA + 4 VALU waves"); run<24, false, false, 6, false, false, true>(out, "roles: 4 MFMA + 4 VALU waves");
  run<32, false, false, 6, false, false, true>(out, "roles: 4 MFMA + 4 VALU waves"); run<32, true, false, 6, false, false, true>(out, "roles + barrier");
  run<16, true, true>(out, "barrier, waves 4-7 V first"); run<24, true, true>(out, "barrier, waves 4-7 V first"); run<32, true, true>(out, "barrier, waves 4-7 V first");
  return 0;
}
