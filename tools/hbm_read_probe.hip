// hbm_read_probe.hip -- what a pure streaming READ sustains on this device, in the two forms the kernels use: LDS-DMA
// (global_load_lds_dwordx4, 1 KB per wave-instruction, no consumer) and plain 16-B-per-lane loads into registers, one
// 256-thread workgroup per CU walking row tiles interleaved over the workgroups (tile w, w + G, ...), D tiles in flight.
// The ceiling the packed weight-gradient kernels (pinn_x6_wgrad.hip) are measured against.
//   hipcc --offload-arch=gfx950 -O3 -o tools/exp/hbm_read_probe tools/hbm_read_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef __attribute__((address_space(1))) const void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int KB, int D>      // KB per tile and workgroup (4 waves x KB/4 pieces), D tiles in flight
__global__ __launch_bounds__(256) void dma_read(const char* src, long long n_tiles, float* out) {
  extern __shared__ __attribute__((aligned(1024))) char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int kPieces = KB / 4;      // per wave and tile
  const long long n_mine = (n_tiles - blockIdx.x + gridDim.x - 1) / gridDim.x;
  auto fetch = [&](long long k, int stage) {
    const char* t = src + (blockIdx.x + (k < n_mine ? k : n_mine - 1) * (long long)gridDim.x) * (KB * 1024LL);
#pragma unroll
    for (int p = 0; p < kPieces; ++p)
      __builtin_amdgcn_global_load_lds((gptr_t)(t + (wave * kPieces + p) * 1024 + lane * 16), (lptr_t)(lds + stage * KB * 1024 + (wave * kPieces + p) * 1024), 16, 0, 0);
  };
#pragma unroll
  for (int s = 0; s < D; ++s) fetch(s, s);
  int st = 0;
  float acc = 0.f;
  for (long long k = 0; k < n_mine; ++k) {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"((D - 1) * kPieces) : "memory");
    acc += *reinterpret_cast<float*>(lds + st * KB * 1024 + threadIdx.x * 4);      // touch the tile (keeps the DMA live)
    __syncthreads();
    fetch(k + D, st);
    st = st + 1 == D ? 0 : st + 1;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  out[blockIdx.x * 256 + threadIdx.x] = acc;
}

template <int U>      // 16-B loads in flight per lane
__global__ __launch_bounds__(256) void reg_read(const f32x4* src, long long n_vec, float* out) {
  f32x4 acc = {0, 0, 0, 0};
  const long long stride = (long long)gridDim.x * 256;
  long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  for (; i + (U - 1) * stride < n_vec; i += U * stride) {
    f32x4 v[U];
#pragma unroll
    for (int u = 0; u < U; ++u) v[u] = src[i + u * stride];
#pragma unroll
    for (int u = 0; u < U; ++u) acc += v[u];
  }
  out[blockIdx.x * 256 + threadIdx.x] = acc[0] + acc[1] + acc[2] + acc[3];
}

int main() {
  const long long bytes = 4LL << 30;
  char* src; float* out;
  if (hipMalloc(&src, bytes) != hipSuccess) return 1;
  (void)hipMalloc(&out, 2048 * 256 * 4);
  (void)hipMemset(src, 0x3c, bytes);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto time = [&](auto launch, const char* name) {
    float best = 1e9f;
    for (int rep = 0; rep < 5; ++rep) {
      (void)hipEventRecord(e0); launch(); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); best = ms < best ? ms : best;
    }
    printf("%-64s %.3f ms  %.2f TB/s\n", name, best, bytes / (best * 1e-3) / 1e12);
  };
  (void)hipFuncSetAttribute((const void*)dma_read<32, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute((const void*)dma_read<16, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
  (void)hipFuncSetAttribute((const void*)dma_read<32, 2>, hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024);
  time([&] { hipLaunchKernelGGL((dma_read<32, 4>), dim3(256), dim3(256), 128 * 1024, 0, src, bytes / (32 * 1024), out); }, "LDS-DMA, 256 WGs, 32-KB tiles, 4 in flight (128 KB / CU)");
  time([&] { hipLaunchKernelGGL((dma_read<32, 2>), dim3(256), dim3(256), 64 * 1024, 0, src, bytes / (32 * 1024), out); }, "LDS-DMA, 256 WGs, 32-KB tiles, 2 in flight (64 KB / CU)");
  time([&] { hipLaunchKernelGGL((dma_read<16, 8>), dim3(256), dim3(256), 128 * 1024, 0, src, bytes / (16 * 1024), out); }, "LDS-DMA, 256 WGs, 16-KB tiles, 8 in flight (128 KB / CU)");
  time([&] { hipLaunchKernelGGL((dma_read<32, 2>), dim3(512), dim3(256), 64 * 1024, 0, src, bytes / (32 * 1024), out); }, "LDS-DMA, 512 WGs (2 / CU), 32-KB tiles, 2 in flight each");
  time([&] { hipLaunchKernelGGL((reg_read<8>), dim3(1024), dim3(256), 0, 0, (const f32x4*)src, bytes / 16, out); }, "16-B loads to registers, 1024 WGs, 8 in flight per lane");
  time([&] { hipLaunchKernelGGL((reg_read<4>), dim3(2048), dim3(256), 0, 0, (const f32x4*)src, bytes / 16, out); }, "16-B loads to registers, 2048 WGs, 4 in flight per lane");
  return 0;
}
