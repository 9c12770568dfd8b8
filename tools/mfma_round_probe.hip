// mfma_round_probe.hip -- how does a 16-bit MFMA round when it adds its products to the fp32 accumulator?
// c = 1 + 2^-23 (an odd last bit) plus a dot product worth f ulps of c, f in {0.25, 0.5, 0.75, 1.25, 1.5, 1.75}, both signs:
// round-to-nearest-even gives {0, +1 (tie to even), +1, +1, +1 ... }; truncation never rounds up in magnitude.
// Prints the accumulator's last bits for v_mfma_f32_16x16x32_f16 / _bf16 and for the exact v_mfma_f32_16x16x4_f32.
// Build: hipcc --offload-arch=gfx950 -O2 tools/mfma_round_probe.hip -o tools/exp/mfma_round_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void probe(float* out, const float* fr, int n) {
  const int lane = threadIdx.x;
  for (int t = 0; t < n; ++t) {
    // A = row of 2^-12 in k = 0 of lane group 0 only, B = fr * 2^-11 ... : one product a*b = fr[t] * 2^-23 (exact in fp16 for these fr)
    f16x8 a = {0, 0, 0, 0, 0, 0, 0, 0}, b = {0, 0, 0, 0, 0, 0, 0, 0};
    bf16x8 ab = {0, 0, 0, 0, 0, 0, 0, 0}, bb = {0, 0, 0, 0, 0, 0, 0, 0};
    if (lane < 16) { a[0] = (_Float16)0x1p-12f; b[0] = (_Float16)(fr[t] * 0x1p-11f); ab[0] = (__bf16)0x1p-12f; bb[0] = (__bf16)(fr[t] * 0x1p-11f); }
    const float c0 = 1.0f + 0x1p-23f;
    f32x4 c = {c0, c0, c0, c0};
    f32x4 r16 = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    f32x4 rbf = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, c, 0, 0, 0);
    f32x4 r32 = __builtin_amdgcn_mfma_f32_16x16x4f32(lane < 16 ? 0x1p-12f : 0.0f, lane < 16 ? fr[t] * 0x1p-11f : 0.0f, c, 0, 0, 0);
    // many small products in ONE instruction: 32 products of fr/32 ulps each (does the sum round once?)
    f16x8 a2, b2;
    for (int j = 0; j < 8; ++j) { a2[j] = (_Float16)0x1p-12f; b2[j] = (_Float16)(fr[t] * 0x1p-16f); }
    f32x4 rs = __builtin_amdgcn_mfma_f32_16x16x32_f16(a2, b2, c, 0, 0, 0);
    if (lane == 0) { out[4 * t + 0] = r16[0]; out[4 * t + 1] = rbf[0]; out[4 * t + 2] = r32[0]; out[4 * t + 3] = rs[0]; }
  }
}

int main() {
  const float fr[12] = {0.25f, 0.5f, 0.75f, 1.25f, 1.5f, 1.75f, -0.25f, -0.5f, -0.75f, -1.25f, -1.5f, -1.75f};
  float *d_out, *d_fr, out[48];
  hipMalloc(&d_out, sizeof(out)); hipMalloc(&d_fr, sizeof(fr));
  hipMemcpy(d_fr, fr, sizeof(fr), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_out, d_fr, 12);
  if (hipMemcpy(out, d_out, sizeof(out), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL\n"); return 1; }
  printf("c = 1 + 1 ulp; result in ulps above 1 (RNE would give: +0.25->1 +0.5->2(tie to even) +0.75->2 +1.25->2 +1.5->2(tie) +1.75->3; -0.25->1 -0.5->0(tie) -0.75->0 -1.25->0 -1.5->0(tie: -0.5 -> 0) -1.75->-1)\n");
  printf("%8s %10s %10s %10s %14s\n", "ulps", "f16 mfma", "bf16 mfma", "f32 mfma", "f16 32 prods");
  for (int t = 0; t < 12; ++t) {
    printf("%8.2f", fr[t]);
    for (int k = 0; k < 4; ++k) printf(" %10.2f", (double)(out[4 * t + k] - 1.0f) / 0x1p-23);
    printf("\n");
  }
  return 0;
}
