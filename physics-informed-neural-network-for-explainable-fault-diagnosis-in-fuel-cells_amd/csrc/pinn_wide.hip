// pinn_wide.hip -- networks wider than the register-resident chain holds (hidden > 256, e.g. BASELINE config 5,
// [8, 1024 x 4, 1]): layer-by-layer kernels on the same building blocks as the x6 chain (pinn_x6_core.h).
//
// Activations live in HBM between layers, in the stash layout [tile16][feature][16 rows] fp32 (scratch behind the
// packed weights in pinn_net_t.d_packed; rows are processed in chunks of kWideChunk).  One layer kernel: a wave owns
// 16 rows and computes 256 (or 128) output features at a time (16 / 8 accumulator blocks); per 32-feature K-group the weight
// slab (the scheme's copies x 256 rows x 64 B) streams global -> LDS by LDS-DMA exactly as in the chain, and the B operand is
// the input activation block (2 KB per wave and group), fetched by LDS-DMA two groups ahead into a per-wave ring and
// split one group ahead, between the MFMA groups of the current slab.  Epilogue per pass:
// bias is in the accumulator; tanh + Philox dropout (the same stream as every other kernel: keyed by global row,
// layer, feature) -> next layer's input, or (backward) times the activation derivative -> d pre-activation.
// Arithmetic as in the fused nets (PINN_PREC_F32X6): forward layers in scheme X3 (two fp16 parts, three MFMAs per product:
// activations x 8 at the split, weights x 64 in the packed copies, the factor 512 leaves in the epilogue), backward layers in
// x6 (three bf16 parts, six MFMAs).
#include "pinn_x6_core.h"

namespace pinn {
namespace wide {

using namespace x6;

constexpr int kWideChunk = 65536;        // rows per pass through the layer kernels (a multiple of 128)

enum { EPI_TANH_DROP = 0, EPI_TANH = 1, EPI_BACKWARD = 2 };

// dropout of one 32-feature group with the mask source chosen at run time (wave-uniform): on-chip Philox, or the bit
// masks a parity test injects (PINN_DROP_BITS: word index (pass * n_rows + row) * words + layer * nb + group, as in the
// fused kernels; the launchers below pre-offset d.bits per row chunk and pass, so pass = 0 here)
__device__ __forceinline__ unsigned activate_pair_rt(f32x4& v0, f32x4& v1, const DropDev& d, const RowCtx& c, const LayerDrop ld, int layer, int fp) {
  return d.mode == PINN_DROP_BITS ? activate_pair<true>(v0, v1, d, c, ld, layer, fp) : activate_pair<false>(v0, v1, d, c, ld, layer, fp);
}

struct LayerArgs {
  const float* params;       // flat fp32 parameters (biases, init weights)
  const char* packed;        // the scheme's copies: three bf16 ones (x6), or the two fp16 ones behind them (X3)
  unsigned copy_bytes;
  const float* in;           // [T16][IN][16]
  float* out;                // [T16][OUT][16]
  const float* act;          // EPI_BACKWARD: post-dropout activation of the OUTPUT features, [T16][OUT][16]
  const float* init_w;       // EPI_BACKWARD, optional: accumulator starts at init_w[f] * init_s[row]
  const float* init_s;
  const float* du;           // EPI_BACKWARD in scheme X3: d loss / d (u, z) per row -> the row's power-of-two normalisation
  const float* dz;
  unsigned* amax;            // EPI_BACKWARD in scheme X3: running max |d pre-activation| of the call (TrainBuffers::amax)
  long long n_rows, row_base;   // rows of this launch; global index of its first row (Philox) = drop.row_offset + chunk start
  int IN, OUT;
  unsigned mat_off;          // bf16-element offset of the [OUT][IN] matrix inside a copy
  int kp_log;                // log2 of OUT: the group stride of the group-major copies is OUT x 32 elements (set by launch_layer)
  long long bias_off;        // float offset of the bias in params (EPI_TANH*)
  int layer;                 // dropout module index
  DropDev drop;
  unsigned pass;
};

// kNT accumulator blocks = 16 kNT output features per pass over the K-groups: 16 (256 features, half the input
// re-reads and fragment splits) when OUT allows, else 8
// kPk (PINN_PREC_F32X6): activations and d pre-activations travel between the kernels as the packed fragments of the fused
// nets' stash (pinn_x6_core.h packed_ptr: per 16-row tile and 32-feature group 2 KB = parts hi, lo of 8 x the activation / of
// the row-normalised gradient) instead of fp32: the consumer's B operand is a 16-B LDS read per part, not a split, and the
// weight gradients run on wgrad_p_kernel.  EPI_TANH (the last variance layer) still writes fp32: only vector sums read it.
template <typename S, int EPI, int kNT, bool kPk = false>
__global__ __launch_bounds__(kThreadsX, 2) void wide_layer_x6_kernel(LayerArgs a) {
  static_assert(!kPk || S::kCopies == 2, "the packed form holds scheme X3's fragments");
  using Pipe = typename S::Pipe;
  using Frag = typename S::Frag;
  constexpr int kOB = 16 * kNT, kNrb = kNT == 16 ? 4 : 3;                 // features per pass; log2(16-row blocks)
  constexpr int kPieces = ((S::kCopies << kNrb) + 7) / 8;                 // LDS-DMA pieces per wave and slab
  static_assert((S::kCopies << kNrb) % 8 == 0, "pieces must divide over the eight waves");
  constexpr int kRingBytes = 8 * 2 * 2048;
  constexpr int kSlabAt = (kRingBytes + 1023) & ~1023;
  constexpr float kAct = S::kActScale, kAcc = S::kAccScale, kInvAcc = 1.0f / S::kAccScale, kWs = S::kAccScale / S::kActScale;
  constexpr bool kNormRows = EPI == EPI_BACKWARD && S::kActScale != 1.0f;      // X3 backward: gradients normalised per row
  float amax = 0.0f;
  __shared__ __attribute__((aligned(1024))) char smem[kSlabAt + 2 * Pipe::kSlab];
  Pipe pipe;
  pipe.lds = smem + kSlabAt;
  pipe.init(a.packed, a.copy_bytes, threadIdx.x);
  const int lane = threadIdx.x & 63, wave = pipe.wave, kq = lane >> 4;
  const StashRing ring{smem + wave * 4096, lane};
  const int NG = a.IN / 32, nob = a.OUT / kOB;
  auto slab_mat = [&](int ob) { return Mat{a.mat_off + (unsigned)(ob * kOB) * 32u, kNrb, 0, a.kp_log}; };      // the rows of pass ob
  // slab (0, 0)
  pipe.par = 0;
  for (int j = 0; j < kPieces; ++j) pipe.template piece<-1>(slab_mat(0), 0, j, 0);
  __syncthreads();

  const long long n_tiles = (a.n_rows + kTileRowsX - 1) / kTileRowsX;
  for (long long tile = blockIdx.x; tile < n_tiles; tile += gridDim.x) {
    const long long t16 = tile * 8 + wave;
    const long long lrow = t16 * 16 + (lane & 15);
    const bool valid = lrow < a.n_rows;
    // (injected masks are indexed by the LOCAL row: padding rows of the last tile read the last real row's words, never past the buffer)
    const RowCtx c{lane, kq, a.row_base + lrow, lrow < a.n_rows ? lrow : a.n_rows - 1, a.n_rows, a.pass, a.drop.mode};
    const float* in_tile = a.in + t16 * a.IN * 16;
    auto fetch = [&](int g) { ring.fetch(in_tile + 32 * g * 16, g & 1); };
    // scale of the B operand: activations x 8 (X3 forward); gradients x the row's normalisation 2^(4 - e), max(|du|, |dz|) =
    // m 2^e (X3 backward, as backward_pass in pinn_x6_core.h); un-normalised again in the epilogue
    float bsc = EPI == EPI_BACKWARD ? 1.0f : kAct, unr = 1.0f;
    if constexpr (kNormRows) {
      const float mxr = valid ? fmaxf(fabsf(a.du[lrow]), fabsf(a.dz[lrow])) : 0.0f;
      int e = 0;
      (void)frexpf(mxr, &e);
      e = mxr > 0.0f ? (e < -120 ? -120 : e) : 4;
      bsc = ldexpf(1.0f, 4 - e);
      unr = ldexpf(1.0f, e - 4);
    }
    // blocks 0 and 1 of this tile's input; block 0 is split at once (nothing to hide it under)
    fetch(0);
    fetch(1);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    Frag cur, nxt;
    if constexpr (kPk) { cur.hi = ring.read_frag(0, 0); cur.lo = ring.read_frag(0, 1); }
    else {
      static_for<4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        S::template split<r>(ring.read(0, 0, r) * bsc, ring.read(0, 1, r) * bsc, cur);
      });
    }
#pragma unroll 1
    for (int ob = 0; ob < nob; ++ob) {
      f32x4 acc[kNT];
      if (EPI == EPI_BACKWARD) {
        const float s = (a.init_w && valid) ? a.init_s[lrow] * (bsc * kWs) : 0.0f;      // padded rows stay exactly zero down the chain
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          const f32x4 w = a.init_w ? *reinterpret_cast<const f32x4*>(a.init_w + ob * kOB + t * 16 + 4 * kq) : f32x4{0.f, 0.f, 0.f, 0.f};
          acc[t] = w * s;
        }
      } else {
        bias_blocks<kNT>(acc, a.params + a.bias_off + ob * kOB, kq);
        if constexpr (kAcc != 1.0f) {
#pragma unroll
          for (int t = 0; t < kNT; ++t) acc[t] = acc[t] * kAcc;
        }
      }
#pragma unroll 1
      for (int g = 0; g < NG; ++g) {
        // next slab, next block to split (g + 1), block to fetch (g + 2); the input blocks repeat for every ob
        const bool last_g = g + 1 == NG;
        const int ob2 = last_g ? (ob + 1 < nob ? ob + 1 : 0) : ob, g2 = last_g ? 0 : g + 1;
        const Mat m2 = slab_mat(ob2);
        const int gb = g2, gf = g + 2 < NG ? g + 2 : g + 2 - NG;
        // slots = tile pairs (kNT / 2): the weight pieces first, the ring fetch in the last one
        auto dma = [&](auto slotc) {
          constexpr int slot = decltype(slotc)::value;
          if constexpr (slot < kPieces) pipe.template piece<-1>(m2, g2, slot, pipe.par ^ 1);
          if constexpr (slot == kNT / 2 - 1) fetch(gf);
        };
        // chunks = tiles: the four register pairs of the next block, evenly spread
        auto vchunk = [&](auto cc) {
          constexpr int ci = decltype(cc)::value, every = kNT / 4;
          if constexpr (kPk) {
            if constexpr (ci == every - 1) { nxt.hi = ring.read_frag(gb & 1, 0); nxt.lo = ring.read_frag(gb & 1, 1); }
          } else if constexpr (ci % every == every - 1) {
            constexpr int r = ci / every;
            S::template split<r>(ring.read(gb & 1, 0, r) * bsc, ring.read(gb & 1, 1, r) * bsc, nxt);
          }
        };
        slab_mfma<S, kNT>(acc, cur, pipe.cur(), lane, vchunk, dma);
        pipe.advance();
        cur = nxt;
      }
      // ---- epilogue of these output features (X3: the accumulators carry 512 x the pre-activation)
      if constexpr (kAcc != 1.0f) {
        const float back = EPI == EPI_BACKWARD ? (kPk ? 1.0f : unr) * (1.0f / kWs) : kInvAcc;
#pragma unroll
        for (int t = 0; t < kNT; ++t) acc[t] = acc[t] * back;
      }
      float* out_tile = a.out + (t16 * a.OUT + ob * kOB + 4 * kq) * 16 + (lane & 15);
      if constexpr (EPI == EPI_BACKWARD && kPk) {
        // act: 8 x the post-dropout activation as (hi, lo); the d pre-activations leave as (hi, lo) in the row's normalised units
        const float scale = a.drop.mode != PINN_DROP_NONE ? a.drop.scale[a.layer] : 1.0f, inv_scale = 1.0f / (scale * kAct);
        const float* hp = packed_ptr(const_cast<float*>(a.act), t16, a.OUT, lane) + (long long)ob * (kNT / 2) * 512;
        float* dp = packed_ptr(a.out, t16, a.OUT, lane) + (long long)ob * (kNT / 2) * 512;
        StashFrag fr[kNT / 2];
#pragma unroll
        for (int k = 0; k < kNT / 2; ++k) {
          fr[k].hi = *reinterpret_cast<const u32x4*>(hp + 512 * k);
          fr[k].lo = *reinterpret_cast<const u32x4*>(hp + 512 * k + 256);
        }
        static_for<kNT / 2>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          Frag out;
          static_for<4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            const float h0 = unpack_act<0, r>(fr[k]), h1 = unpack_act<1, r>(fr[k]);
            const float a0 = h0 * inv_scale, a1 = h1 * inv_scale;
            const float g0 = acc[2 * k][r] * (scale * (1.0f - a0 * a0)), g1 = acc[2 * k + 1][r] * (scale * (1.0f - a1 * a1));
            const float p0 = h0 != 0.0f ? g0 : 0.0f, p1 = h1 != 0.0f ? g1 : 0.0f;
            amax = fmaxf(amax, fmaxf(fabsf(p0), fabsf(p1)) * unr);
            S::template split<r>(p0, p1, out);
          });
          PINN_STASH_ST(reinterpret_cast<u32x4*>(dp + 512 * k), out.hi);
          PINN_STASH_ST(reinterpret_cast<u32x4*>(dp + 512 * k + 256), out.lo);
        });
      } else if (EPI == EPI_BACKWARD) {
        const float scale = a.drop.mode != PINN_DROP_NONE ? a.drop.scale[a.layer] : 1.0f, inv_scale = 1.0f / scale;
        const float* hp = a.act + (t16 * a.OUT + ob * kOB + 4 * kq) * 16 + (lane & 15);
        f32x4 hl[kNT];
#pragma unroll
        for (int t = 0; t < kNT; ++t) load_block(hp, t, hl[t]);
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float av = hl[t][r] * inv_scale;
            const float gv = acc[t][r] * (scale * (1.0f - av * av));
            acc[t][r] = hl[t][r] != 0.0f ? gv : 0.0f;
            if constexpr (kNormRows) amax = fmaxf(amax, fabsf(acc[t][r]));
          }
          store_block(out_tile, t, acc[t]);
        }
      } else if constexpr (EPI == EPI_TANH_DROP && kPk) {
        const LayerDrop ldr = layer_drop(a.drop, c.mode, a.layer);
        float* op = packed_ptr(a.out, t16, a.OUT, lane) + (long long)ob * (kNT / 2) * 512;
        static_for<kNT / 2>([&](auto kc) {
          constexpr int k = decltype(kc)::value;
          const unsigned keep = activate_pair_rt(acc[2 * k], acc[2 * k + 1], a.drop, c, ldr, a.layer, ob * (kNT / 2) + k);
          Frag out;
          static_for<4>([&](auto rc) {
            constexpr int r = decltype(rc)::value;
            // 8 x the activation (exact); a kept activation that is exactly 0 becomes the smallest fp16 subnormal ("dropped" is h == 0)
            float h0 = acc[2 * k][r] * kAct, h1 = acc[2 * k + 1][r] * kAct;
            h0 = (((keep >> r) & 1u) && h0 == 0.0f) ? 0x1p-24f : h0;
            h1 = (((keep >> (4 + r)) & 1u) && h1 == 0.0f) ? 0x1p-24f : h1;
            S::template split<r>(h0, h1, out);
          });
          PINN_STASH_ST(reinterpret_cast<u32x4*>(op + 512 * k), out.hi);
          PINN_STASH_ST(reinterpret_cast<u32x4*>(op + 512 * k + 256), out.lo);
        });
      } else if (EPI == EPI_TANH_DROP) {
        const LayerDrop ldr = layer_drop(a.drop, c.mode, a.layer);
#pragma unroll
        for (int k = 0; k < kNT / 2; ++k) {
          const unsigned keep = activate_pair_rt(acc[2 * k], acc[2 * k + 1], a.drop, c, ldr, a.layer, ob * (kNT / 2) + k);
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            acc[2 * k][r] = stash_value(acc[2 * k][r], (keep >> r) & 1u);
            acc[2 * k + 1][r] = stash_value(acc[2 * k + 1][r], (keep >> (4 + r)) & 1u);
          }
          store_block(out_tile, 2 * k, acc[2 * k]);
          store_block(out_tile, 2 * k + 1, acc[2 * k + 1]);
        }
      } else {
#pragma unroll
        for (int t = 0; t < kNT; ++t) {
          activate_tanh(acc[t]);
          store_block(out_tile, t, acc[t]);
        }
      }
    }
  }
  if constexpr (kNormRows) {          // the call's largest |d pre-activation|: the fp16 weight-gradient kernels' common scale
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) atomicMax(a.amax, __float_as_uint(amax));
  }
}

// input layer: h0 = dropout(tanh(W0 x + b0)) -> stash; 8 MACs per output, VALU
struct InputArgs {
  const float* params;
  const float* x;
  float* out;               // [T16][H][16]
  long long n_rows, row_base;
  int H;
  long long w0_off, b0_off;
  DropDev drop;
  unsigned pass;
  int packed;               // PINN_PREC_F32X6: `out` as packed fragments (wide_layer_x6_kernel kPk)
  float* stash_x;           // packed && training: the input rows as a packed group of their own (x / 16: layer 0's weight gradient), else nullptr
};
__global__ __launch_bounds__(256) void wide_input_kernel(InputArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4;
  const long long n_t16 = (a.n_rows + 127) / 128 * 8;
  const LayerDrop ldr = layer_drop(a.drop, a.drop.mode, 0);
  for (long long t16 = (long long)blockIdx.x * 4 + wave; t16 < n_t16; t16 += (long long)gridDim.x * 4) {
    const long long lrow = t16 * 16 + (lane & 15);
    const long long srow = lrow < a.n_rows ? lrow : a.n_rows - 1;
    const f32x4 xa = reinterpret_cast<const f32x4*>(a.x)[srow * 2], xb = reinterpret_cast<const f32x4*>(a.x)[srow * 2 + 1];
    // (injected masks are indexed by the LOCAL row: padding rows of the last tile read the last real row's words, never past the buffer)
    const RowCtx c{lane, kq, a.row_base + lrow, lrow < a.n_rows ? lrow : a.n_rows - 1, a.n_rows, a.pass, a.drop.mode};
    float* out_tile = a.out + (t16 * a.H + 4 * kq) * 16 + (lane & 15);
    float* out_pk = packed_ptr(a.out, t16, a.H, lane);
    if (a.stash_x) {
      Frag2 fx;
      static_for<4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        const float v = kq == 0 ? xa[r] : (kq == 1 ? xb[r] : 0.0f);
        X3::split<r>(v * 0.0625f, 0.0f, fx);
      });
      float* xp = packed_ptr(a.stash_x, t16, 32, lane);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(xp), fx.hi);
      PINN_STASH_ST(reinterpret_cast<u32x4*>(xp + 256), fx.lo);
    }
    for (int fp = 0; fp < a.H / 32; ++fp) {
      f32x4 v[2];
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int f = 32 * fp + 16 * b + 4 * kq + r;
          const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.params + a.w0_off + f * 8);
          const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.params + a.w0_off + f * 8 + 4);
          float s = a.params[a.b0_off + f];
          // the k order of the fused kernels' input MFMA (k-step t contracts inputs {t, 4 + t}) is immaterial here: plain fp32 FMAs
          s = fmaf(w0[0], xa[0], s); s = fmaf(w0[1], xa[1], s); s = fmaf(w0[2], xa[2], s); s = fmaf(w0[3], xa[3], s);
          s = fmaf(w1[0], xb[0], s); s = fmaf(w1[1], xb[1], s); s = fmaf(w1[2], xb[2], s); s = fmaf(w1[3], xb[3], s);
          v[b][r] = s;
        }
      const unsigned keep = activate_pair_rt(v[0], v[1], a.drop, c, ldr, 0, fp);
      if (a.packed) {
        Frag2 out;
        static_for<4>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          float h0 = v[0][r] * X3::kActScale, h1 = v[1][r] * X3::kActScale;
          h0 = (((keep >> r) & 1u) && h0 == 0.0f) ? 0x1p-24f : h0;
          h1 = (((keep >> (4 + r)) & 1u) && h1 == 0.0f) ? 0x1p-24f : h1;
          X3::split<r>(h0, h1, out);
        });
        PINN_STASH_ST(reinterpret_cast<u32x4*>(out_pk + 512 * fp), out.hi);
        PINN_STASH_ST(reinterpret_cast<u32x4*>(out_pk + 512 * fp + 256), out.lo);
        continue;
      }
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        v[0][r] = stash_value(v[0][r], (keep >> r) & 1u);
        v[1][r] = stash_value(v[1][r], (keep >> (4 + r)) & 1u);
      }
      store_block(out_tile, 2 * fp, v[0]);
      store_block(out_tile, 2 * fp + 1, v[1]);
    }
  }
}

// heads: u = w_p . h_last + b_p, z = wv_2 . v2 + bv_2, logvar = log(softplus(z) + 1e-6)
//   mode 0: write (u, logvar);  1: MC eval pass (pred_mean = u, reset the sums);  2: MC stochastic pass (accumulate)
struct HeadArgs {
  const float* params;
  const float* h;           // [T16][H][16]
  const float* v2;          // [T16][H/4][16]
  long long n_rows;
  int H;
  long long wp_off, bp_off, wv2_off, bv2_off;
  int mode;
  float* o0; float* o1;     // mode 0: u, logvar; mode 1: pred_mean
  float* accum;             // modes 1, 2: [4][chunk rows]: u_eval, running mean of du, Welford m2 of du, sum logvar
  long long accum_stride;
  int pass;                 // mode 2: 0-based stochastic pass index
  int packed;               // h as packed fragments
};
// <w_p, h> over this lane's share of the last hidden layer (all lanes of a row add up in sum_kq)
__device__ __forceinline__ float head_dot(const float* h, long long t16, int H, int lane, const float* wp, int packed) {
  const int kq = lane >> 4;
  float up = 0.f;
  if (packed) {
    const float* hp = packed_ptr(const_cast<float*>(h), t16, H, lane);
    for (int g = 0; g < H / 32; ++g) {
      StashFrag fr;
      fr.hi = *reinterpret_cast<const u32x4*>(hp + 512 * g);
      fr.lo = *reinterpret_cast<const u32x4*>(hp + 512 * g + 256);
      const f32x4 w0 = *reinterpret_cast<const f32x4*>(wp + 32 * g + 4 * kq), w1 = *reinterpret_cast<const f32x4*>(wp + 32 * g + 16 + 4 * kq);
      static_for<4>([&](auto rc) {
        constexpr int r = decltype(rc)::value;
        up = fmaf(w0[r], unpack_act<0, r>(fr) * (1.0f / X3::kActScale), up);
        up = fmaf(w1[r], unpack_act<1, r>(fr) * (1.0f / X3::kActScale), up);
      });
    }
  } else {
    const float* hp = h + (t16 * H + 4 * kq) * 16 + (lane & 15);
    for (int t = 0; t < H / 16; ++t) {
      f32x4 hv;
      load_block(hp, t, hv);
      up = block_dot(hv, wp + t * 16, kq, up);
    }
  }
  return up;
}
__global__ __launch_bounds__(256) void wide_heads_kernel(HeadArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4;
  const long long n_t16 = (a.n_rows + 127) / 128 * 8;
  for (long long t16 = (long long)blockIdx.x * 4 + wave; t16 < n_t16; t16 += (long long)gridDim.x * 4) {
    const long long lrow = t16 * 16 + (lane & 15);
    const float* vp = a.v2 + (t16 * (a.H / 4) + 4 * kq) * 16 + (lane & 15);
    float up = head_dot(a.h, t16, a.H, lane, a.params + a.wp_off, a.packed), zp = 0.f;
    for (int t = 0; t < a.H / 64; ++t) {
      f32x4 vv;
      load_block(vp, t, vv);
      zp = block_dot(vv, a.params + a.wv2_off + t * 16, kq, zp);
    }
    const float u = sum_kq(up) + a.params[a.bp_off], z = sum_kq(zp) + a.params[a.bv2_off];
    const float lv = logf(softplus_f32(z) + 1e-6f);
    if (lrow < a.n_rows && lane < 16) {
      if (a.mode == 0) { a.o0[lrow] = u; a.o1[lrow] = lv; }
      else if (a.mode == 1) {
        a.o0[lrow] = u;
        a.accum[lrow] = u; a.accum[a.accum_stride + lrow] = 0.f; a.accum[2 * a.accum_stride + lrow] = 0.f; a.accum[3 * a.accum_stride + lrow] = 0.f;
      } else {
        float mean = a.accum[a.accum_stride + lrow], m2 = a.accum[2 * a.accum_stride + lrow];
        welford_update(mean, m2, u - a.accum[lrow], 1.0f / (float)(a.pass + 1));
        a.accum[a.accum_stride + lrow] = mean;
        a.accum[2 * a.accum_stride + lrow] = m2;
        a.accum[3 * a.accum_stride + lrow] += lv;
      }
    }
  }
}
__global__ __launch_bounds__(256) void wide_mc_finalize_kernel(const float* accum, long long stride, long long n, int n_passes, float* a_u, float* e_u) {
  const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float inv_t = 1.0f / (float)n_passes;
  const float var = accum[2 * stride + i] * inv_t;          // Welford m2 / T
  a_u[i] = expf(0.5f * (accum[3 * stride + i] * inv_t));
  e_u[i] = sqrtf(var);
}

// heads + aleatoric_loss (01:916-927) + its gradient for the training chain: du, dz per row, loss partial sums per
// workgroup, and d pre_v2 = wv2 * dz * (1 - v2^2) into the stash
struct LossArgs {
  const float* params;
  const float* h;           // [T16][H][16]  last hidden layer
  const float* v2;          // [T16][H/4][16]
  float* dv2;               // [T16][H/4][16]
  const float* y;
  float* du; float* dz;     // [T16 * 16]
  double* loss_part;        // [grid][8]
  long long n_rows, n_global;
  int H;
  long long wp_off, bp_off, wv2_off, bv2_off;
  unsigned* amax;            // running max |d pre-activation| of the call (this kernel: d pre_v2), or nullptr
  int packed;                // PINN_PREC_F32X6: h and dv2 as packed fragments, dv2 in the rows' normalised units
  unsigned* emax;            // packed: the call's largest max(|du|, |dz|) (zeroed before the launch): the row scales' reference
};
__global__ __launch_bounds__(256) void wide_loss_kernel(LossArgs a) {
  float amax = 0.0f, gmax = 0.0f;
  __shared__ double red[4][8];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, kq = lane >> 4;
  const long long n_t16 = (a.n_rows + 127) / 128 * 8;
  const float inv_n = (float)(1.0 / (double)a.n_global);
  float s_nll = 0.f, s_abs = 0.f, s_mse = 0.f, s_du = 0.f, s_dz = 0.f;
  for (long long t16 = (long long)blockIdx.x * 4 + wave; t16 < n_t16; t16 += (long long)gridDim.x * 4) {
    const long long lrow = t16 * 16 + (lane & 15);
    const bool valid = lrow < a.n_rows;
    const float* vp = a.v2 + (t16 * (a.H / 4) + 4 * kq) * 16 + (lane & 15);
    float up = head_dot(a.h, t16, a.H, lane, a.params + a.wp_off, a.packed), zp = 0.f;
    for (int t = 0; t < a.H / 64; ++t) {
      f32x4 vv;
      load_block(vp, t, vv);
      zp = block_dot(vv, a.params + a.wv2_off + t * 16, kq, zp);
    }
    const float u = sum_kq(up) + a.params[a.bp_off], z = sum_kq(zp) + a.params[a.bv2_off];
    const float yv = a.y[valid ? lrow : a.n_rows - 1];
    float du = 0.f, dz = 0.f;
    {
      const float sp = softplus_f32(z);
      const float var = sp + 1e-6f;
      const float s = logf(var);
      const float prec = expf(-s);
      const float e = yv - u;
      if (valid) {
        du = -(prec * e) * inv_n;
        const float sgn = (s > 0.f) ? 1.f : ((s < 0.f) ? -1.f : 0.f);
        const float ds = (-0.5f * prec * e * e + 0.5f + 0.01f * sgn) * inv_n;
        const float sig = z > 20.0f ? 1.0f : 1.0f / (1.0f + expf(-z));
        dz = ds * sig / var;
        if (kq == 0) {
          s_nll += 0.5f * prec * e * e + 0.5f * s;
          s_abs += fabsf(s);
          s_mse += e * e;
          s_du += du;
          s_dz += dz;
        }
      }
      if (lane < 16) { a.du[t16 * 16 + lane] = du; a.dz[t16 * 16 + lane] = dz; }
    }
    if (a.packed) {
      // the backward layers' first operand in the row's normalised units (norm = 2^(4 - e), max(|du|, |dz|) = m 2^e), as (hi, lo)
      gmax = fmaxf(gmax, fmaxf(fabsf(du), fabsf(dz)));
      const int e = grad_exponent(fmaxf(fabsf(du), fabsf(dz)), 4);
      const float norm = ldexpf(1.0f, 4 - e), unr = ldexpf(1.0f, e - 4), dzn = dz * norm;
      float* dpk = packed_ptr(a.dv2, t16, a.H / 4, lane);
      for (int g = 0; g < a.H / 128; ++g) {
        f32x4 v0, v1;
        load_block(vp, 2 * g, v0);
        load_block(vp, 2 * g + 1, v1);
        const f32x4 w0 = *reinterpret_cast<const f32x4*>(a.params + a.wv2_off + 32 * g + 4 * kq);
        const f32x4 w1 = *reinterpret_cast<const f32x4*>(a.params + a.wv2_off + 32 * g + 16 + 4 * kq);
        Frag2 out;
        static_for<4>([&](auto rc) {
          constexpr int r = decltype(rc)::value;
          const float p0 = w0[r] * dzn * (1.0f - v0[r] * v0[r]), p1 = w1[r] * dzn * (1.0f - v1[r] * v1[r]);
          amax = fmaxf(amax, fmaxf(fabsf(p0), fabsf(p1)) * unr);
          X3::split<r>(p0, p1, out);
        });
        PINN_STASH_ST(reinterpret_cast<u32x4*>(dpk + 512 * g), out.hi);
        PINN_STASH_ST(reinterpret_cast<u32x4*>(dpk + 512 * g + 256), out.lo);
      }
      continue;
    }
    float* dp = a.dv2 + (t16 * (a.H / 4) + 4 * kq) * 16 + (lane & 15);
    for (int t = 0; t < a.H / 64; ++t) {
      f32x4 vv;
      load_block(vp, t, vv);
      const f32x4 w = *reinterpret_cast<const f32x4*>(a.params + a.wv2_off + t * 16 + 4 * kq);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        vv[r] = w[r] * dz * (1.0f - vv[r] * vv[r]);
        amax = fmaxf(amax, fabsf(vv[r]));
      }
      store_block(dp, t, vv);
    }
  }
  if (a.amax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) amax = fmaxf(amax, __shfl_xor(amax, off, 64));
    if (lane == 0) atomicMax(a.amax, __float_as_uint(amax));
  }
  if (a.packed && a.emax) {
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, off, 64));
    if (lane == 0) atomicMax(a.emax, __float_as_uint(gmax));
  }
  float terms[5] = {s_nll, s_abs, s_mse, s_du, s_dz};
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    double v = (double)terms[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    if (lane == 0) red[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < 8) {
    double t = 0.0;
    if (threadIdx.x < 5)
      for (int w = 0; w < 4; ++w) t += red[w][threadIdx.x];
    a.loss_part[(long long)blockIdx.x * 8 + threadIdx.x] = t;
  }
}

// The packed weight gradients' row records (pinn_x6_core.h RowMeta): per 16-row tile fp16 t_r = 2^(e_r - E + c), the two fp16 parts of
// du_r norm_r, fp32 dz_r.  E comes from the loss kernel's maximum, complete when this launch starts.
__global__ __launch_bounds__(256) void wide_rowmeta_kernel(const float* __restrict__ du, const float* __restrict__ dz, const unsigned* __restrict__ emax,
                                                           int qboost, _Float16* __restrict__ meta, long long t16_total) {
  const int E = grad_exponent(__uint_as_float(*emax), 4);
  const long long row = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= t16_total * 16) return;
  const long long t16 = row >> 4;
  const int n = (int)(row & 15);
  const float d_u = du[row], d_z = dz[row];
  const int e = grad_exponent(fmaxf(fabsf(d_u), fabsf(d_z)), E);
  const float dun = d_u * ldexpf(1.0f, 4 - e);
  _Float16* rec = meta + t16 * 128;
  const _Float16 dh16 = (_Float16)dun;
  rec[n] = (_Float16)ldexpf(1.0f, e - E + qboost);
  rec[16 + n] = dh16;
  rec[32 + n] = (_Float16)(dun - (float)dh16);
  reinterpret_cast<float*>(rec)[32 + n] = d_z;
}

}  // namespace wide

namespace x6 {
void launch_pack_x6(const pinn_net_t* net, const float* d_params, hipStream_t st, unsigned* zero_word = nullptr);   // pinn_x6.hip
}

// scratch behind the packed weights (floats): two activation buffers, v1, v2, the MC sums
// PINN_PREC_F32X6: scheme X3 on the fp16 copies (behind the three bf16 ones), forward and backward (_G6: backward layers in x6);
// PINN_PREC_BF16: one bf16 part per operand everywhere (the first bf16 copy)
template <int EPI>
static void launch_wide_layer(wide::LayerArgs la, int grid, hipStream_t st, int precision) {
  la.kp_log = 31 - __builtin_clz((unsigned)la.OUT);      // group-major weight copies: group stride = OUT x 32 elements
  if (precision == PINN_PREC_BF16) {
    if (la.OUT % 256 == 0) hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::B1, EPI, 16>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
    else hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::B1, EPI, 8>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
  } else if (EPI == wide::EPI_BACKWARD && precision == PINN_PREC_F32X6_G6) {
    if (la.OUT % 256 == 0) hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X6, EPI, 16>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
    else hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X6, EPI, 8>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
  } else if (precision == PINN_PREC_F32X6) {      // packed activations / gradients between the kernels
    la.packed += 3 * (size_t)la.copy_bytes;
    if (la.OUT % 256 == 0) hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X3, EPI, 16, true>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
    else hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X3, EPI, 8, true>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
  } else {
    la.packed += 3 * (size_t)la.copy_bytes;
    if (la.OUT % 256 == 0) hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X3, EPI, 16>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
    else hipLaunchKernelGGL((wide::wide_layer_x6_kernel<x6::X3, EPI, 8>), dim3(grid), dim3(x6::kThreadsX), 0, st, la);
  }
}

size_t wide_scratch_floats(int H) {
  return (size_t)wide::kWideChunk * (2 * (size_t)H + H / 2 + H / 4 + 4);
}

static int wide_cus() { return cu_count_cached(); }

// forward / MC-dropout of a wide net: chunks of rows through input -> hidden layers -> variance head -> heads
int launch_forward_wide(const pinn_net_t* net, const FwdArgs& fa, bool mc, void* stream) {
  using namespace wide;
  hipStream_t st = (hipStream_t)stream;
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  PackLayout K{H, nh};
  x6::launch_pack_x6(net, fa.params, st);
  const char* packed = (const char*)net->d_packed;
  const unsigned copy_bytes = (unsigned)(K.total() * 2);
  float* scratch = (float*)((char*)net->d_packed + (size_t)K.total() * 2 * 5);
  float* bufA = scratch; float* bufB = bufA + (size_t)kWideChunk * H;
  float* v1 = bufB + (size_t)kWideChunk * H; float* v2 = v1 + (size_t)kWideChunk * (H / 2);
  float* accum = v2 + (size_t)kWideChunk * (H / 4);
  const int cus = wide_cus();
  const int pk = net->precision == PINN_PREC_F32X6;      // activations between the kernels as packed fragments
  const int n_passes = mc ? fa.n_passes : 0;
  for (long long r0 = 0; r0 < fa.n_rows; r0 += kWideChunk) {
    const long long n = fa.n_rows - r0 < kWideChunk ? fa.n_rows - r0 : kWideChunk;
    const long long tiles = (n + 127) / 128;
    const int grid_l = (int)(tiles < cus ? tiles : cus), grid_s = (int)((tiles * 2 < 4 * cus) ? tiles * 2 : 4 * cus);
    for (int pass = -1; pass < n_passes; ++pass) {
      DropDev d = fa.drop;
      const bool stochastic = mc ? pass >= 0 : fa.drop.mode != PINN_DROP_NONE;
      if (!stochastic) d.mode = PINN_DROP_NONE;
      unsigned p = pass < 0 ? 0u : (unsigned)pass;
      if (d.mode == PINN_DROP_BITS) {      // the masks of this (pass, row chunk): kernels index them from pass 0, row 0
        d.bits = fa.drop.bits + ((long long)p * fa.n_rows + r0) * d.words;
        p = 0u;
      }
      InputArgs ia{fa.params, fa.x + r0 * 8, bufA, n, d.row_offset + r0, H, L.w0(), L.b0(), d, p, pk, nullptr};
      hipLaunchKernelGGL(wide_input_kernel, dim3(grid_s), dim3(256), 0, st, ia);
      float* cur = bufA; float* nxt = bufB;
      LayerArgs la{};
      la.params = fa.params; la.packed = packed; la.copy_bytes = copy_bytes; la.n_rows = n; la.row_base = d.row_offset + r0; la.drop = d; la.pass = p;
      for (int l = 1; l < nh; ++l) {
        la.in = cur; la.out = nxt; la.IN = H; la.OUT = H; la.mat_off = (unsigned)K.w(l); la.kp_log = 31 - __builtin_clz((unsigned)H);
        la.bias_off = L.b(l); la.layer = l;
        launch_wide_layer<EPI_TANH_DROP>(la, grid_l, st, net->precision);
        float* t = cur; cur = nxt; nxt = t;
      }
      la.in = cur; la.out = v1; la.IN = H; la.OUT = H / 2; la.mat_off = (unsigned)K.wv0(); la.kp_log = 31 - __builtin_clz((unsigned)H);
      la.bias_off = L.bv0(); la.layer = nh;
      launch_wide_layer<EPI_TANH_DROP>(la, grid_l, st, net->precision);
      la.in = v1; la.out = v2; la.IN = H / 2; la.OUT = H / 4; la.mat_off = (unsigned)K.wv1(); la.kp_log = 31 - __builtin_clz((unsigned)round_up64(H / 2));
      la.bias_off = L.bv1(); la.layer = nh + 1;
      launch_wide_layer<EPI_TANH>(la, grid_l, st, net->precision);
      HeadArgs ha{fa.params, cur, v2, n, H, L.wp(), L.bp(), L.wv2(), L.bv2(), mc ? (pass < 0 ? 1 : 2) : 0,
                  fa.o0 + r0, mc ? nullptr : fa.o1 + r0, accum, kWideChunk, pass < 0 ? 0 : pass, pk};
      hipLaunchKernelGGL(wide_heads_kernel, dim3(grid_s), dim3(256), 0, st, ha);
      if (!mc) break;
    }
    if (mc) hipLaunchKernelGGL(wide_mc_finalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, accum, (long long)kWideChunk, n, n_passes,
                               fa.o1 + r0, fa.o2 + r0);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

// chain phase of pinn_mlp_train_grads for a wide net: every layer's activation and d pre-activation to the stash (the
// layout the weight-gradient kernels read); *grid_out = number of loss partials
int launch_train_chain_wide(const pinn_net_t* net, const float* d_params, const float* d_x, const float* d_y, long long n_rows,
                            long long n_global, const DropDev& drop, const TrainBuffers& b, int* grid_out, void* stream) {
  using namespace wide;
  hipStream_t st = (hipStream_t)stream;
  const int H = net->hidden, nh = net->n_hidden;
  ParamLayout L{H, nh};
  PackLayout K{H, nh};
  x6::launch_pack_x6(net, d_params, st);
  const char* packed = (const char*)net->d_packed;
  const unsigned copy_bytes = (unsigned)(K.total() * 2);
  const int cus = wide_cus();
  const long long tiles = (n_rows + 127) / 128;
  const int grid_l = (int)(tiles < cus ? tiles : cus), grid_s = (int)((tiles * 2 < 4 * cus) ? tiles * 2 : 4 * cus);
  const long long hs = b.t16 * H * 16;            // floats per hidden-layer stash
  float* sh = (float*)b.stash_h; float* sv1 = (float*)b.stash_v1; float* sv2 = (float*)b.stash_v2;
  float* dh = (float*)b.dpre_h; float* dv1 = (float*)b.dpre_v1; float* dv2 = (float*)b.dpre_v2;
  auto log2i = [](int v) { return 31 - __builtin_clz((unsigned)v); };

  const int pk = net->precision == PINN_PREC_F32X6;      // packed stash: the weight gradients run on wgrad_p_kernel (pinn_train.hip)
  InputArgs ia{d_params, d_x, sh, n_rows, drop.row_offset, H, L.w0(), L.b0(), drop, 0u, pk, pk ? (float*)b.stash_x : nullptr};
  hipLaunchKernelGGL(wide_input_kernel, dim3(grid_s), dim3(256), 0, st, ia);
  LayerArgs la{};
  la.params = d_params; la.packed = packed; la.copy_bytes = copy_bytes; la.n_rows = n_rows; la.row_base = drop.row_offset; la.drop = drop; la.pass = 0;
  for (int l = 1; l < nh; ++l) {
    la.in = sh + (l - 1) * hs; la.out = sh + l * hs; la.IN = H; la.OUT = H; la.mat_off = (unsigned)K.w(l); la.kp_log = log2i(H);
    la.bias_off = L.b(l); la.layer = l;
    launch_wide_layer<EPI_TANH_DROP>(la, grid_l, st, net->precision);
  }
  la.in = sh + (nh - 1) * hs; la.out = sv1; la.IN = H; la.OUT = H / 2; la.mat_off = (unsigned)K.wv0(); la.kp_log = log2i(H);
  la.bias_off = L.bv0(); la.layer = nh;
  launch_wide_layer<EPI_TANH_DROP>(la, grid_l, st, net->precision);
  la.in = sv1; la.out = sv2; la.IN = H / 2; la.OUT = H / 4; la.mat_off = (unsigned)K.wv1(); la.kp_log = log2i(round_up64(H / 2));
  la.bias_off = L.bv1(); la.layer = nh + 1;
  launch_wide_layer<EPI_TANH>(la, grid_l, st, net->precision);

  const long long t4 = b.t16 / 4;
  const int grid_loss = (int)(t4 < 1024 ? (t4 < 1 ? 1 : t4) : 1024);
  const bool x3_grads = net->precision == PINN_PREC_F32X6;          // gradients in scheme X3: the call's largest |d pre| is recorded
  if (x3_grads) {
    hipError_t em = hipMemsetAsync(b.amax, 0, 2 * sizeof(unsigned), st);      // amax and emax (TrainBuffers: consecutive words)
    if (em != hipSuccess) return (int)em;
  }
  LossArgs lo{d_params, sh + (nh - 1) * hs, sv2, dv2, d_y, b.du, b.dz, b.loss_part, n_rows, n_global, H, L.wp(), L.bp(), L.wv2(), L.bv2(),
              x3_grads ? b.amax : nullptr, pk, pk ? b.emax : nullptr};
  hipLaunchKernelGGL(wide_loss_kernel, dim3(grid_loss), dim3(256), 0, st, lo);
  if (pk) hipLaunchKernelGGL(wide_rowmeta_kernel, dim3((unsigned)((b.t16 * 16 + 255) / 256)), dim3(256), 0, st, b.du, b.dz, b.emax, b.qboost,
                             (_Float16*)b.rowmeta, b.t16);
  la.du = b.du; la.dz = b.dz; la.amax = b.amax;
  *grid_out = grid_loss;

  // backward: Wv1^T, Wv0^T (+ w_p du), W_l^T for l = nh-1 .. 1
  la.init_w = nullptr; la.init_s = nullptr;
  la.in = dv2; la.out = dv1; la.act = sv1; la.IN = H / 4; la.OUT = H / 2; la.mat_off = (unsigned)K.wv1t(); la.kp_log = log2i(round_up64(H / 4));
  la.layer = nh;
  launch_wide_layer<EPI_BACKWARD>(la, grid_l, st, net->precision);
  la.in = dv1; la.out = dh + (nh - 1) * hs; la.act = sh + (nh - 1) * hs; la.IN = H / 2; la.OUT = H; la.mat_off = (unsigned)K.wv0t();
  la.kp_log = log2i(round_up64(H / 2)); la.layer = nh - 1; la.init_w = d_params + L.wp(); la.init_s = b.du;
  launch_wide_layer<EPI_BACKWARD>(la, grid_l, st, net->precision);
  la.init_w = nullptr; la.init_s = nullptr;
  for (int l = nh - 1; l >= 1; --l) {
    la.in = dh + l * hs; la.out = dh + (l - 1) * hs; la.act = sh + (l - 1) * hs; la.IN = H; la.OUT = H; la.mat_off = (unsigned)K.wt(l);
    la.kp_log = log2i(H); la.layer = l - 1;
    launch_wide_layer<EPI_BACKWARD>(la, grid_l, st, net->precision);
  }
  hipError_t e = hipGetLastError();
  return e == hipSuccess ? PINN_OK : (int)e;
}

}  // namespace pinn
