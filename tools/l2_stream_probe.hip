// l2_stream_probe.hip -- how fast ONE workgroup per CU can pull an L2-resident stream into its LDS: the chain kernels' weight
// stream at small row counts (every workgroup walks the same 0.7 MB of packed weights slab by slab; 32 KB per slab).
// Forms: LDS-DMA (buffer/global_load ... lds, 1 KB per wave-instruction) with 1 or 3 slabs of lookahead, and register-staged
// (global_load_dwordx4 -> ds_write_b128) -- 4 or 8 waves per workgroup, 64 / 157 / 256 workgroups.  Prints ns per 32-KB slab.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/l2_stream_probe tools/l2_stream_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int kSlab = 32 * 1024, kSlabs = 22;      // 0.69 MB, walked cyclically

template <int WAVES, int D>      // D slabs in LDS; D - 1 requested ahead
__global__ __launch_bounds__(WAVES * 64) void dma_stream(const char* src, int n_steps, float* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kPieces = 32 / WAVES;
  auto fetch = [&](int k, int stage) {
    const char* t = src + (k % kSlabs) * kSlab;
#pragma unroll
    for (int p = 0; p < kPieces; ++p)
      __builtin_amdgcn_global_load_lds((gptr_t)(t + (wave * kPieces + p) * 1024 + lane * 16), (lptr_t)(lds + stage * kSlab + (wave * kPieces + p) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < D - 1; ++s) fetch(s, s);
  int st = 0;
  float acc = 0.f;
  for (int k = 0; k < n_steps; ++k) {
    const int fill = st + D - 1 >= D ? st - 1 : st + D - 1;
    fetch(k + D - 1, fill);
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * kPieces) : "memory");
    __syncthreads();
    acc += *reinterpret_cast<float*>(lds + st * kSlab + threadIdx.x * 4);
    __syncthreads();
    st = st + 1 == D ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int WAVES>      // register-staged: the loads of slab k + 1 in flight while slab k is written to LDS and "used"
__global__ __launch_bounds__(WAVES * 64) void reg_stream(const char* src, int n_steps, float* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  constexpr int kVec = kSlab / 16 / (WAVES * 64);      // 16-B vectors per thread and slab
  f32x4 v[kVec];
  auto load = [&](int k) {
    const f32x4* t = reinterpret_cast<const f32x4*>(src + (k % kSlabs) * kSlab);
#pragma unroll
    for (int u = 0; u < kVec; ++u) v[u] = t[u * WAVES * 64 + threadIdx.x];
  };
  load(0);
  int st = 0;
  float acc = 0.f;
  for (int k = 0; k < n_steps; ++k) {
    f32x4* d = reinterpret_cast<f32x4*>(lds + st * kSlab);
#pragma unroll
    for (int u = 0; u < kVec; ++u) d[u * WAVES * 64 + threadIdx.x] = v[u];
    load(k + 1);
    __syncthreads();
    acc += *reinterpret_cast<float*>(lds + st * kSlab + threadIdx.x * 4);
    __syncthreads();
    st ^= 1;
  }
  out[blockIdx.x * 512 + threadIdx.x] = acc + v[0][0];
}

int main() {
  char* src; float* out;
  if (hipMalloc(&src, kSlab * kSlabs) != hipSuccess) return 1;
  (void)hipMalloc(&out, 512 * 512 * 4);
  (void)hipMemset(src, 0x3c, kSlab * kSlabs);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  const int n_steps = 20000;
  auto time = [&](auto launch, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-72s %7.1f ns per 32-KB slab  (%.1f GB/s per CU)\n", name, best * 1e6 / n_steps, kSlab / (best * 1e-3 / n_steps) / 1e9);
  };
#define ATTR(K, B) (void)hipFuncSetAttribute((const void*)K, hipFuncAttributeMaxDynamicSharedMemorySize, B)
  ATTR((dma_stream<4, 2>), 2 * kSlab); ATTR((dma_stream<8, 2>), 2 * kSlab); ATTR((dma_stream<8, 4>), 4 * kSlab); ATTR((dma_stream<4, 4>), 4 * kSlab);
  ATTR((reg_stream<4>), 2 * kSlab); ATTR((reg_stream<8>), 2 * kSlab);
  for (int g : {64, 157, 256}) {
    printf("-- %d workgroups\n", g);
    time([&] { hipLaunchKernelGGL((dma_stream<4, 2>), dim3(g), dim3(256), 2 * kSlab, 0, src, n_steps, out); }, "LDS-DMA, 4 waves, 1 slab ahead");
    time([&] { hipLaunchKernelGGL((dma_stream<8, 2>), dim3(g), dim3(512), 2 * kSlab, 0, src, n_steps, out); }, "LDS-DMA, 8 waves, 1 slab ahead");
    time([&] { hipLaunchKernelGGL((dma_stream<4, 4>), dim3(g), dim3(256), 4 * kSlab, 0, src, n_steps, out); }, "LDS-DMA, 4 waves, 3 slabs ahead");
    time([&] { hipLaunchKernelGGL((dma_stream<8, 4>), dim3(g), dim3(512), 4 * kSlab, 0, src, n_steps, out); }, "LDS-DMA, 8 waves, 3 slabs ahead");
    time([&] { hipLaunchKernelGGL((reg_stream<4>), dim3(g), dim3(256), 2 * kSlab, 0, src, n_steps, out); }, "registers -> ds_write_b128, 4 waves (8 x 16 B per lane in flight)");
    time([&] { hipLaunchKernelGGL((reg_stream<8>), dim3(g), dim3(512), 2 * kSlab, 0, src, n_steps, out); }, "registers -> ds_write_b128, 8 waves (4 x 16 B per lane in flight)");
  }
  return 0;
}
