// clock_probe.hip -- the bare matrix-instruction stream of the X3 forward kernels with in-kernel clock stamps (see run()).
// Derived from x3_slab_bench.hip:
// What would a 2-part fp16 split (x = hi + lo with 11 + 11 significand bits; products hi.hi + hi.lo + lo.hi: three
// v_mfma_f32_16x16x32_f16 per product instead of six bf16 ones) buy the multiply phase?  Same shape as x6_slab_bench
// (512-thread workgroup = 2 waves per SIMD x 16 rows, 16 output tiles per 32-feature K-group, A fragments from LDS, the
// next slab by LDS-DMA, one barrier per slab) with 2 weight copies / 2 ds_read_b128 / 3 MFMAs per tile, beside the x6
// shape (3 / 3 / 6) in the same binary: ns per slab step under the chip's power management, not cycles.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I physics-*/csrc -I include -o x3_slab_bench tools/x3_slab_bench.hip
#include "pinn_x6_core.h"
#include <cstdio>
using namespace pinn;
using namespace pinn::x6;

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ unsigned long long g_stamps[256][4];
template <int NPARTS>     // 3: x6 (bf16, 6 MFMAs per tile);  2: fp16 hi/lo (3 MFMAs per tile)
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
  u32x4 b[3];
  for (int p = 0; p < 3; ++p) for (int q = 0; q < 4; ++q) b[p][q] = 0x3c003800u + 17u * lane + 1000u * q + 77u * p;    // fp16 / bf16 bit patterns ~1
  const Mat m{0u, 4};
  const int kq = lane >> 4, i = lane & 15;
  unsigned long long t0 = __builtin_readcyclecounter();
  { unsigned long long c, r; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
    if (threadIdx.x == 0) { g_stamps[blockIdx.x][0] = c; g_stamps[blockIdx.x][1] = r; } }
  for (int it = 0; it < iters; ++it) {
    const char* base = pipe.cur() + i * 64 + ((kq ^ swz(i)) << 4);
    const unsigned addr = (unsigned)(unsigned long long)(lptr_t)base;
    AFrag3 a0, a1;
    __builtin_amdgcn_sched_barrier(0);
    load_a3<0>(a0, addr); load_a3<1>(a1, addr);
    __builtin_amdgcn_sched_barrier(0);
    static_for<16>([&](auto tc) {
      constexpr int t = decltype(tc)::value;
      AFrag3& a = (t & 1) ? a1 : a0;
      if constexpr (t + 2 < 16) wait_a3<3>(a); else wait_a3<0>(a);
      f32x4 c = acc[t];
      if constexpr (NPARTS == 3) {
        const Frag3 bb{b[0], b[1], b[2]};
        mfma6(c, a, bb);
      } else {
        const f16x8 ah = __builtin_bit_cast(f16x8, a.h), al = __builtin_bit_cast(f16x8, a.m);
        const f16x8 bh = __builtin_bit_cast(f16x8, b[0]), bl = __builtin_bit_cast(f16x8, b[1]);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, c, 0, 0, 0);
        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, c, 0, 0, 0);
      }
      acc[t] = c;
      __builtin_amdgcn_sched_barrier(0);
      if constexpr (t + 2 < 16) {
        // NPARTS == 2: only two copies are read (the third read is dropped)
        if constexpr (NPARTS == 3) load_a3<t + 2>(a, addr);
        else { a.h = lds_read_b128<(t + 2) * 1024>(addr); a.m = lds_read_b128<kSlabBytes / 3 + (t + 2) * 1024>(addr); a.l = a.m; }
      }
      if constexpr ((t & 1) == 0) { constexpr int slot = t / 2; if (slot < 2 * NPARTS) pipe.piece<8>(m, it & 7, slot, pipe.par ^ 1); }
      __builtin_amdgcn_sched_barrier(0);
    });
    __syncthreads(); pipe.par ^= 1;
  }
  unsigned long long t1 = __builtin_readcyclecounter();
  { unsigned long long c, r; asm volatile("s_memtime %0\n\ts_memrealtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(c), "=s"(r) :: "memory");
    if (threadIdx.x == 0) { g_stamps[blockIdx.x][2] = c; g_stamps[blockIdx.x][3] = r; } }
  float s = 0;
  for (int t = 0; t < 16; ++t) s += acc[t][0] + acc[t][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

#include <algorithm>
#include <chrono>
#include <vector>
// the guide's recipe ("DVFS give-back" item 6): >= 2 s of back-to-back launches on random operands, then the in-kernel clock
// = delta s_memtime / delta s_memrealtime x 100 MHz around the loop, median over workgroups
template <int NPARTS>
void run(float* out, __bf16* packed, unsigned long long* cyc) {
  const int iters = 20000;
  auto w0 = std::chrono::steady_clock::now();
  while (std::chrono::duration<double>(std::chrono::steady_clock::now() - w0).count() < 2.5) {
    hipLaunchKernelGGL(k<NPARTS>, dim3(256), dim3(512), 0, 0, out, packed, iters, cyc);
    (void)hipDeviceSynchronize();
  }
  float ms = 0;
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k<NPARTS>, dim3(256), dim3(512), 0, 0, out, packed, iters, cyc);
  (void)hipEventRecord(e1); (void)hipEventSynchronize(e1); (void)hipEventElapsedTime(&ms, e0, e1);
  unsigned long long h[256][4];
  (void)hipMemcpyFromSymbol(h, HIP_SYMBOL(g_stamps), sizeof(h));
  std::vector<double> clk;
  for (int b = 0; b < 256; ++b) clk.push_back((double)(h[b][2] - h[b][0]) / (double)(h[b][3] - h[b][1]) * 0.1);
  std::sort(clk.begin(), clk.end());
  const double ns = ms * 1e6 / iters;
  printf("bare stream, %d parts (%d MFMAs per tile, two waves per SIMD, LDS reads + weight DMA + barrier, no activation work): %.0f ns per slab step, "
         "in-kernel clock median %.3f GHz (min %.3f, max %.3f) -> %.0f cycles per step, matrix pipe %.1f %% busy\n",
         NPARTS, NPARTS == 3 ? 6 : 3, ns, clk[128], clk[0], clk[255], ns * clk[128], 100.0 * (NPARTS == 3 ? 3072 : 1536) / (ns * clk[128]));
}

int main() {
  float* out; __bf16* packed; unsigned long long* cyc;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&packed, 8 << 20); (void)hipMalloc(&cyc, 8);
  // random operands (zeros would flatter the clock)
  std::vector<unsigned short> hst(4 << 20);
  unsigned s = 12345u;
  for (auto& v : hst) { s = s * 1664525u + 1013904223u; v = (unsigned short)(0x3400u + ((s >> 16) & 0x0fffu) + ((s >> 3) & 0x8000u)); }      // fp16 ~ +-[0.25, 1)
  (void)hipMemcpy(packed, hst.data(), 8 << 20, hipMemcpyHostToDevice);
  run<2>(out, packed, cyc);
  return 0;
}
