// tr16_probe.hip -- exact-integer check of the two instructions the packed weight-gradient kernel leans on (gfx950):
//   (1) ds_read_b64_tr_b16: which LDS element lands in which lane / element of the result, for the image
//       [16 rows][4 kq][8 x f16] (one stash piece of 1 KB) addressed as wgrad_p_kernel addresses it;
//   (2) v_dot2_f32_f16 (__builtin_amdgcn_fdot2): f32 accumulation, fp16 subnormal operands honoured or flushed.
// Build: hipcc --offload-arch=gfx950 -O2 tools/tr16_probe.hip -o tools/exp/tr16_probe ; prints PASS / FAIL lines.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) void* lptr_t;

__global__ void probe(const _Float16* src, _Float16* out, float* dots) {
  __shared__ __attribute__((aligned(1024))) _Float16 img[16 * 32];
  for (int e = threadIdx.x; e < 512; e += 64) img[e] = src[e];
  __syncthreads();
  const int lane = threadIdx.x, hh = lane >> 5, x = (lane >> 4) & 1, L = lane & 15, q = L >> 2, p = L & 3;
  for (int m = 0; m < 2; ++m) {
    // lane 4q + p of a 16-lane group supplies the address of (row q of the block, 8-byte piece p)
    const unsigned addr = (unsigned)(unsigned long long)(lptr_t)img + (8 * hh + 4 * m + q) * 64 + (2 * x + (p >> 1)) * 16 + (p & 1) * 8;
    f16x4 v;
    asm volatile("ds_read_b64_tr_b16 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    for (int e = 0; e < 4; ++e) out[(lane * 2 + m) * 4 + e] = v[e];
  }
  if (lane == 0) {
    const f16x2 a = {(_Float16)3.0f, (_Float16)5.0f}, b = {(_Float16)7.0f, (_Float16)11.0f};
    dots[0] = __builtin_amdgcn_fdot2(a, b, 0.5f, false);                        // 21 + 55 + 0.5
    const f16x2 sa = {(_Float16)5.9604645e-8f, (_Float16)0.0f}, sb = {(_Float16)16384.0f, (_Float16)0.0f};   // 2^-24 * 2^14 = 2^-10
    dots[1] = __builtin_amdgcn_fdot2(sa, sb, 0.0f, false);
    const f16x2 la = {(_Float16)60000.0f, (_Float16)60000.0f}, lb = {(_Float16)60000.0f, (_Float16)60000.0f};
    dots[2] = __builtin_amdgcn_fdot2(la, lb, 1.0f, false);                      // 7.2e9 + 1: products beyond fp16 range, f32 sum
    const f16x2 ta = {(_Float16)1.0f, (_Float16)0.0009765625f}, tb = {(_Float16)1.0f, (_Float16)0.0009765625f};
    dots[3] = __builtin_amdgcn_fdot2(ta, tb, 0.0f, false);                      // 1 + 2^-20: exact in f32
  }
}

int main() {
  _Float16 h[512];
  for (int r = 0; r < 16; ++r) for (int c = 0; c < 32; ++c) h[r * 32 + c] = (_Float16)(float)(r * 32 + c);
  _Float16 *d_src, *d_out; float* d_dots;
  hipMalloc(&d_src, sizeof(h)); hipMalloc(&d_out, 64 * 8 * 2); hipMalloc(&d_dots, 16);
  hipMemcpy(d_src, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, d_src, d_out, d_dots);
  _Float16 o[64 * 8]; float dots[4];
  if (hipMemcpy(o, d_out, sizeof(o), hipMemcpyDeviceToHost) != hipSuccess) { printf("FAIL: launch\n"); return 1; }
  hipMemcpy(dots, d_dots, sizeof(dots), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int lane = 0; lane < 64; ++lane)
    for (int m = 0; m < 2; ++m)
      for (int e = 0; e < 4; ++e) {
        const int hh = lane >> 5, x = (lane >> 4) & 1, L = lane & 15, pc = L >> 2, el = L & 3;
        const int row = 8 * hh + 4 * m + e, col = 8 * (2 * x + (pc >> 1)) + 4 * (pc & 1) + el;
        const float want = (float)(row * 32 + col), got = (float)o[(lane * 2 + m) * 4 + e];
        if (want != got) { if (bad < 8) printf("  lane %d m %d e %d: want row %d col %d (%g) got %g (row %d col %d)\n", lane, m, e, row, col, want, got, (int)got / 32, (int)got % 32); ++bad; }
      }
  printf("%s: ds_read_b64_tr_b16 mapping (%d mismatches)\n", bad ? "FAIL" : "PASS", bad);
  printf("%s: fdot2 basic = %g (want 76.5)\n", dots[0] == 76.5f ? "PASS" : "FAIL", dots[0]);
  printf("%s: fdot2 subnormal operand = %g (want %g)\n", dots[1] == 0.0009765625f ? "PASS" : "FAIL", dots[1], 0.0009765625);
  printf("%s: fdot2 large products = %.10g (want 7200000001 ~ 7.2e9)\n", fabsf(dots[2] - 7.2e9f) < 1e3f ? "PASS" : "FAIL", dots[2]);
  printf("%s: fdot2 f32 sum = %.10g (want %.10g)\n", dots[3] == 1.0f + 0x1p-20f ? "PASS" : "FAIL", dots[3], 1.0 + 0x1p-20);
  return bad != 0;
}
