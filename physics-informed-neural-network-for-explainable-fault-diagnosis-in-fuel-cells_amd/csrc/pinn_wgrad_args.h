// pinn_wgrad_args.h -- arguments of the weight-gradient kernels (exact-fp32: pinn_train.hip; split-bf16: pinn_x6_wgrad.hip)
#pragma once

namespace pinn {

// dW[i][j] = sum_rows P[i][row] Q[j][row]  (+ bias / vector sums), operands in the tiled stash layout
struct WgradArgs {
  const float* P;    // [T16][OUT][16]   d pre-activation of this layer
  const float* Q;    // [T16][IN][16]    its input activation (or nullptr: read x rows, IN = 8)
  const float* x;    // [n_rows][8] when Q == nullptr
  long long n_rows;
  int OUT, IN;
  long long t16;
  int n_slices;
  long long slab_stride;    // floats between consecutive slices' slabs (= padded param count)
  float* dW;                // slab of slice 0: [OUT][IN] row-major at the parameter's offset
  float* db;                // slab of slice 0: [OUT]
  const float* s1; float* dvq;                    // optional: dvq[j] = sum_rows s1[row] Q[j][row]
  const float* s2; const float* R; float* dvr;    // optional: dvr[i] = sum_rows s2[row] R[i][row]
  const unsigned* amax;     // fp16 scheme only: bits of max |P| over the call (TrainBuffers::amax)
};

// the packed form (PINN_PREC_F32X6 on the fused nets): operands are the chain kernels' fp16 fragments, pinn_x6_core.h packed_ptr
struct WgradPArgs {
  const char* P;     // [T16][OUT / 32][2 KB]  d pre-activation in the rows' normalised units: parts hi, lo
  const char* Q;     // [T16][IN / 32][2 KB]   8 x the input activation: parts hi, lo
  const void* meta;  // [T16][256 B]: struct RowMeta records (row scales t_r, du_r norm_r as two fp16 parts, dz_r)
  const unsigned* emax;   // bits of the call's largest max(|du|, |dz|) -> E
  int qboost;             // c
  int q_log2;             // log2 of the scale the Q stash carries: 3 (8 x the activation) or -4 (the input rows, x / 16)
  int ldW, n_cols;        // 0, or (layer 0: IN = 32 of which 8 are real) the leading dimension of dW and the columns to write
  int OUT, IN;
  long long t16;
  int n_slices;
  long long slab_stride;
  float* dW; float* db;
  float* dvq;                                     // optional: dvq[j] = sum_rows du[row] Q[j][row]   (du from the meta records)
  const float* s2; const float* R; float* dvr;    // optional: dvr[i] = sum_rows dz[row] R[i][row]   (R fp32 [T16][OUT][16]; dz from the meta records, s2 only says so)
};

// several packed weight-gradient problems in one launch (small row counts, H = 256: pinn_x6_wgrad.hip wgrad_p_multi_kernel)
constexpr int kMaxWgradProblems = 12;
struct WgradPMulti {
  WgradPArgs p[kMaxWgradProblems];
  int kind[kMaxWgradProblems];        // 0 layer 0, 1 hidden [256][256], 2 variance head 0 (+ predict weight), 3 variance head 1 (+ last weight)
  int first[kMaxWgradProblems + 1];   // (filled by the launcher) first workgroup of problem k
  int n;
};

}  // namespace pinn
