// Results assembly on the device: the float64 [N, 22] `comprehensive_results` array of
// create_comprehensive_results_array_v2 (reference 01:1877-2010) in one pass over the rows.
// 84 B/row read (rows, target, three MC outputs, eight residual columns, label), 176 B/row written;
// one workgroup = 256 consecutive rows,
// the smoothing windows of the two uncertainty columns staged in LDS.
#include <hip/hip_runtime.h>

#include "../../include/pinn_hip.h"

namespace pinn {
namespace {

constexpr int kRows = 256;              // rows per workgroup = threads
constexpr int kMaxWindow = 1024;        // smoothing window limit (reference: 200)

struct ResultsDev {
  double x_min[8], x_scale[8];
  double y_min, y_scale;                // sklearn inverse_transform of the float32 targets (01:1917)
  double mc_min, mc_div;                // pred = (pred_mean - mc_min) / mc_div, std = s / mc_div; mc_div = scale_y + 1e-12 (01:1933-1936)
  int half, right;                      // pandas centred window: rows [i - half, i + right] (01:1830-1845)
  int n_seg;
};

__device__ __forceinline__ double denorm32(float v, double mn, double sc) {
  // numpy in-place `X -= min_; X /= scale_` on a float32 array with float64 operands, then widened to float64
  const float t = (float)((double)v - mn);
  return (double)(float)((double)t / sc);
}

// seg_end: exclusive ends of the smoothing segments, ascending, last == n_rows (01:1848-1872)
__global__ __launch_bounds__(kRows) void results_kernel(const float* __restrict__ x, const float* __restrict__ y, ResultsDev a,
                                                        const long long* __restrict__ seg_end, const float* __restrict__ pm,
                                                        const float* __restrict__ au, const float* __restrict__ eu,
                                                        const float* __restrict__ cols, long long ld,
                                                        const float* __restrict__ labels, long long n_rows, double* __restrict__ out) {
  __shared__ double s_au[kRows + kMaxWindow], s_eu[kRows + kMaxWindow];
  __shared__ __attribute__((aligned(16))) double s_out[kRows * 22];
  const long long row0 = (long long)blockIdx.x * kRows;
  const long long lo = row0 - a.half > 0 ? row0 - a.half : 0;
  const long long hi_want = row0 + kRows + a.right;
  const long long hi = hi_want < n_rows ? hi_want : n_rows;          // staged rows [lo, hi)
  for (long long j = lo + threadIdx.x; j < hi; j += kRows) {
    s_au[j - lo] = (double)au[j] / a.mc_div;
    s_eu[j - lo] = (double)eu[j] / a.mc_div;
  }
  __syncthreads();
  // every thread builds its row in an LDS image of the workgroup's [256, 22] output tile (contiguous in memory), which
  // then goes out as full 16-B-per-lane stores; a thread writing its own 176-B row would scatter 8-B stores
  const long long i = row0 + threadIdx.x;
  if (i < n_rows) {
    // this row's segment [s0, s1)
    long long s0 = 0, s1 = n_rows;
    for (int k = 0; k < a.n_seg; ++k) {
      const long long e = seg_end[k];
      if (i < e) { s1 = e; break; }
      s0 = e;
    }
    long long ws = i - a.half, we = i + a.right + 1;
    ws = ws > s0 ? ws : s0;
    we = we < s1 ? we : s1;
    double sa = 0.0, se = 0.0;
    for (long long j = ws; j < we; ++j) { sa += s_au[j - lo]; se += s_eu[j - lo]; }
    const double cnt = (double)(we - ws);

    double* o = s_out + threadIdx.x * 22;
    const float4 xa = reinterpret_cast<const float4*>(x)[2 * i], xb = reinterpret_cast<const float4*>(x)[2 * i + 1];
    const float xv[8] = {xa.x, xa.y, xa.z, xa.w, xb.x, xb.y, xb.z, xb.w};
#pragma unroll
    for (int c = 0; c < 8; ++c) o[c] = denorm32(xv[c], a.x_min[c], a.x_scale[c]);
    const double yt = denorm32(y[i], a.y_min, a.y_scale);
    const double pred = ((double)pm[i] - a.mc_min) / a.mc_div;
    o[8] = yt;
    o[9] = pred;
    o[10] = sa / cnt;
    o[11] = se / cnt;
    o[12] = yt - pred;
    o[13] = (double)cols[PINN_C_FV * ld + i];
    o[14] = (double)cols[PINN_C_FT * ld + i];
    o[15] = (double)cols[PINN_C_FH * ld + i];
    o[16] = (double)cols[PINN_C_FO * ld + i];
    o[17] = labels ? (double)labels[i] : 0.0;
    o[18] = (double)cols[PINN_C_VEST5 * ld + i];
    o[19] = (double)cols[PINN_C_TPRED * ld + i];
    o[20] = (double)cols[PINN_C_ACTH * ld + i];
    o[21] = (double)cols[PINN_C_ACTO * ld + i];
  }
  __syncthreads();
  const long long n_valid = n_rows - row0 < kRows ? n_rows - row0 : kRows;
  typedef double f64x2 __attribute__((ext_vector_type(2)));
  f64x2* dst = reinterpret_cast<f64x2*>(out + row0 * 22);
  const f64x2* src = reinterpret_cast<const f64x2*>(s_out);
  for (int k = threadIdx.x; k < n_valid * 11; k += kRows) dst[k] = src[k];
}

}  // namespace
}  // namespace pinn

extern "C" int pinn_results_assemble(const float* d_x, const float* d_y, const pinn_affine_t* aff, double mc_min, double mc_scale,
                                     int window, const long long* d_seg_end, int n_segments, const float* d_pred_mean,
                                     const float* d_a_u, const float* d_e_u, const float* d_cols, long long ld,
                                     const float* d_labels, long long n_rows, double* d_out, void* stream) {
  using namespace pinn;
  if (n_rows < 0 || !aff || window < 1 || window > kMaxWindow || n_segments < 0 || (n_segments > 0 && !d_seg_end)) return PINN_E_ARG;
  if (n_rows == 0) return PINN_OK;
  if (!d_x || !d_y || !d_pred_mean || !d_a_u || !d_e_u || !d_cols || !d_out || ld < n_rows) return PINN_E_ARG;
  if (((unsigned long long)d_out | (unsigned long long)d_x) & 15) return PINN_E_ARG;      // 16-B vector accesses
  ResultsDev a;
  for (int c = 0; c < 8; ++c) { a.x_min[c] = aff->x_min[c]; a.x_scale[c] = aff->x_scale[c]; }
  a.y_min = aff->y_min; a.y_scale = aff->y_scale;
  a.mc_min = mc_min; a.mc_div = mc_scale + 1e-12;
  a.half = window / 2;
  a.right = window % 2 == 0 ? a.half - 1 : a.half;
  a.n_seg = n_segments;
  (void)hipGetLastError();
  const unsigned blocks = (unsigned)((n_rows + kRows - 1) / kRows);
  hipLaunchKernelGGL(results_kernel, dim3(blocks), dim3(kRows), 0, (hipStream_t)stream, d_x, d_y, a, d_seg_end, d_pred_mean, d_a_u, d_e_u,
                     d_cols, ld, d_labels, n_rows, d_out);
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}
