// pinn_optim.hip -- torch.optim.Adam (defaults) over the flat parameter vector (01:939, 954).
//
//   m = lerp(m, g, 1-b1); v = b2 v + (1-b2) g g; p -= lr/(1-b1^t) * m / (sqrt(v)/sqrt(1-b2^t) + eps)
// Elementwise, HBM-bound on 4 streams of n floats (n = 175 k for the reference net): one launch.
#include <hip/hip_runtime.h>
#include <math.h>
#include "../../include/pinn_hip.h"
#include "pinn_adam_update.h"

namespace {
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                   float* __restrict__ v, long long n, float step_size, float bc2_sqrt) {
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    pinn::adam_update(pi, g[i], mi, vi, step_size, bc2_sqrt);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}
// the same update with (step_size, bc2_sqrt) taken from a device table at the device's own step count (replayed graphs)
__global__ __launch_bounds__(256) void adam_dev_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m,
                                                       float* __restrict__ v, long long n, const float* __restrict__ coeffs,
                                                       const unsigned* __restrict__ step_counter) {
  const unsigned k = *step_counter - 1u;
  const float step_size = coeffs[2 * k], bc2_sqrt = coeffs[2 * k + 1];
  const long long stride = (long long)gridDim.x * blockDim.x;
  for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
    float pi = p[i], mi = m[i], vi = v[i];
    pinn::adam_update(pi, g[i], mi, vi, step_size, bc2_sqrt);
    p[i] = pi;
    m[i] = mi;
    v[i] = vi;
  }
}
}  // namespace

extern "C" void pinn_adam_coeffs(float lr, int step, float* step_size, float* bc2_sqrt) {
  const double bc1 = 1.0 - pow(0.9, (double)step);
  const double bc2 = 1.0 - pow(0.999, (double)step);
  *step_size = (float)((double)lr / bc1);
  *bc2_sqrt = (float)sqrt(bc2);
}

extern "C" int pinn_adam_step_dev(float* d_params, const float* d_grads, float* d_m, float* d_v, long long n, const float* d_coeffs,
                                  const unsigned* d_step_counter, void* stream) {
  if (!d_params || !d_grads || !d_m || !d_v || !d_coeffs || !d_step_counter || n < 0) return PINN_E_ARG;
  if (n == 0) return PINN_OK;
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  (void)hipGetLastError();
  hipLaunchKernelGGL(adam_dev_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_params, d_grads, d_m, d_v, n, d_coeffs,
                     d_step_counter);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

extern "C" int pinn_adam_step(float* d_params, const float* d_grads, float* d_m, float* d_v, long long n, float lr, int step,
                              void* stream) {
  if (!d_params || !d_grads || !d_m || !d_v || n < 0 || step < 1) return PINN_E_ARG;
  if (n == 0) return PINN_OK;
  float step_size, bc2_sqrt;
  pinn_adam_coeffs(lr, step, &step_size, &bc2_sqrt);
  long long blocks = (n + 255) / 256;
  if (blocks > 2048) blocks = 2048;
  (void)hipGetLastError();   // drop a stale error left by another HIP user of this thread
  hipLaunchKernelGGL(adam_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, d_params, d_grads, d_m, d_v, n,
                     step_size, bc2_sqrt);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}
