// HBM-bound NHWC passes around the convolutions: train/eval BatchNorm (+dropout-aware backward),
// activation, residual add, HRNet multi-resolution fuse, bilinear up-sampling into the 960-channel
// concat buffer, bias gradients, global average pool.
// Reference semantics: lib/nn/modules/batchnorm.py:56-61 (F.batch_norm fallback),
// models/hrnetv2_nodownsp.py:46-64 (BasicBlock), :228-252 (fuse), :434-442 (concat),
// models/model_utils.py:254-255 (AvgPool2d(10)).
#include "common.h"
#include <stdlib.h>

namespace {

// Column-group layout shared by the BN kernels: a thread owns one float4 channel group
// (cw = C/4 groups) and walks rows with stride rpi = 256/cw.
struct RowWalk {
  int cw, rpi, col, r0;
  __device__ RowWalk(int C) {
    cw = C >> 2;
    rpi = 256 / cw; if (rpi < 1) rpi = 1;
    col = threadIdx.x % cw;
    r0 = threadIdx.x / cw;
  }
  __device__ bool active() const { return r0 < rpi; }
};

// ---- BN statistics: per-channel sum and sum of squares (double accumulators) --------------
// All BN kernels work on a channel window [c0, c0+C) of rows with leading dimension ld (C <= 1024 per
// launch; wider layers are covered by several launches); per-channel pointers arrive offset by c0.
__global__ __launch_bounds__(256) void bn_stats_kernel(const float* __restrict__ y, long M, int C, int ld, int Ctot,
                                                       int rows_per_block, double* __restrict__ sums /* [block][2][Ctot] + c0 */) {
  extern __shared__ double sm[];   // [rpi][C][2] partials
  RowWalk w(C);
  const long rb = (long)blockIdx.x * rows_per_block;
  long re = rb + rows_per_block; if (re > M) re = M;
  f32x4 s = {0, 0, 0, 0}, ss = {0, 0, 0, 0};
  if (w.active())
    for (long r = rb + w.r0; r < re; r += w.rpi) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(y + r * ld + 4 * w.col);
      s += v; ss += v * v;
    }
  if (w.active()) {
    double* p = sm + ((long)w.r0 * C + 4 * w.col) * 2;
    p[0] = s.x; p[1] = ss.x; p[2] = s.y; p[3] = ss.y; p[4] = s.z; p[5] = ss.z; p[6] = s.w; p[7] = ss.w;
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    double a = 0, b = 0;
    for (int r = 0; r < w.rpi; ++r) { a += sm[((long)r * C + c) * 2]; b += sm[((long)r * C + c) * 2 + 1]; }
    sums[(long)blockIdx.x * 2 * Ctot + c] = a;          // this block's record: summed in block order by bn_finalize_kernel
    sums[(long)blockIdx.x * 2 * Ctot + Ctot + c] = b;
  }
}

// ---- BN finalize: mean / invstd, running-stat update ---------------------------------------
__global__ void bn_finalize_kernel(const double* __restrict__ sums, int nblk, long M, int C, float momentum, float eps,
                                   float* __restrict__ running_mean, float* __restrict__ running_var,
                                   float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < nblk; ++k) { s1 += sums[(long)k * 2 * C + c]; s2 += sums[(long)k * 2 * C + C + c]; }
  const double mu = s1 / (double)M;
  double var = s2 / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const float unbiased = (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    running_mean[c] = momentum * (float)mu + (1.f - momentum) * running_mean[c];
    running_var[c] = momentum * unbiased + (1.f - momentum) * running_var[c];
  }
}

// one block per channel: reduce the per-workgroup partials written by the conv epilogue (double tree),
// then the same finalisation as bn_finalize_kernel
__global__ __launch_bounds__(256) void bn_finalize_slab_kernel(const float* __restrict__ slab, int nwg, long M, int C,
                                                               float momentum, float eps, float* __restrict__ running_mean,
                                                               float* __restrict__ running_var, float* __restrict__ mean,
                                                               float* __restrict__ invstd) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0, ss = 0.0;
  for (int i = threadIdx.x; i < nwg; i += blockDim.x) {
    s += (double)slab[((long)i * C + c) * 2];
    ss += (double)slab[((long)i * C + c) * 2 + 1];
  }
  s = block_sum<double>(s, red);
  ss = block_sum<double>(ss, red);
  if (threadIdx.x != 0) return;
  const double mu = s / (double)M;
  double var = ss / (double)M - mu * mu;
  if (var < 0) var = 0;
  mean[c] = (float)mu;
  invstd[c] = (float)(1.0 / sqrt(var + (double)eps));
  if (running_mean != nullptr) {
    const float unbiased = (float)(var * ((double)M / (double)(M > 1 ? M - 1 : 1)));
    running_mean[c] = momentum * (float)mu + (1.f - momentum) * running_mean[c];
    running_var[c] = momentum * unbiased + (1.f - momentum) * running_var[c];
  }
}

// eval-mode BatchNorm as one multiply-add per element: scale = gamma / sqrt(var + eps), shift = beta - mean * scale
__global__ void bn_eval_affine_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                      const float* __restrict__ gamma, const float* __restrict__ beta, int C, float eps,
                                      float* __restrict__ scale, float* __restrict__ shift) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float sc = gamma[c] * (1.0f / sqrtf(running_var[c] + eps));
  scale[c] = sc;
  shift[c] = beta[c] - running_mean[c] * sc;
}

__global__ void bn_eval_prepare_kernel(const float* __restrict__ running_mean, const float* __restrict__ running_var,
                                       int C, float eps, float* __restrict__ mean, float* __restrict__ invstd) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  mean[c] = running_mean[c];
  invstd[c] = 1.0f / sqrtf(running_var[c] + eps);
}

// ---- BN apply + residual + activation --------------------------------------------------------
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const float* __restrict__ y, const float* __restrict__ mean,
                                                         const float* __restrict__ invstd, const float* __restrict__ gamma,
                                                         const float* __restrict__ beta, const float* __restrict__ res,
                                                         float* __restrict__ out, unsigned char* __restrict__ mask, long M, int C,
                                                         int ld, int rows_per_block, int act) {
  RowWalk w(C);
  if (!w.active()) return;
  const int c = 4 * w.col;
  f32x4 sc, sh;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float a = invstd[c + j] * gamma[c + j];
    sc[j] = a; sh[j] = beta[c + j] - mean[c + j] * a;
  }
  const long rb = (long)blockIdx.x * rows_per_block;
  long re = rb + rows_per_block; if (re > M) re = M;
  for (long r = rb + w.r0; r < re; r += w.rpi) {
    f32x4 v = *reinterpret_cast<const f32x4*>(y + r * ld + c);
    v = v * sc + sh;
    if (res != nullptr) v += *reinterpret_cast<const f32x4*>(res + r * ld + c);
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = fs_act(v[j], act);
    *reinterpret_cast<f32x4*>(out + r * ld + c) = v;
    if (mask != nullptr) {      // activation-derivative bits of the 4 channels, one byte per float4 (read back by the backward passes)
      unsigned m = 0u;
#pragma unroll
      for (int j = 0; j < 4; ++j) m |= (fs_act_mask(v[j], act) != 0.f ? 1u : 0u) << j;
      mask[(r * ld + c) >> 2] = (unsigned char)m;
    }
  }
}

// ---- BN backward -------------------------------------------------------------------------------
// Three stages, none of which touches the matrix cores, LDS beyond 8 KB or more than 48 VGPRs -- so that their workgroups fit NEXT TO
// the convolution kernels of the other HRNet branches (two 230-VGPR waves per SIMD and 154 of the CU's 160 KB of LDS leave exactly
// that much), instead of waiting for a CU to drain:
//   partial : per-block column sums  S = sum(g), SX = sum(g * xhat),  g = dz * act'(z),  xhat = (y - mean) * invstd  -> slab[blk][C][2]
//             (skipped when the kernel that PRODUCED dz already wrote the slab: add_n_bnsum_kernel below)
//   finalize: slab -> dgamma, dbeta and the per-channel coefficients of the apply pass
//   apply   : dy = ga * g - d * (y - mean) - bb   [x dropout mask],  dres = g
//             with ga = gamma*invstd, d = ga * invstd * SX/M, bb = ga * S/M   (eval mode: d = bb = 0)
// Algebra as F.batch_norm's backward: dy = ga * (g - mean(g) - xhat * mean(g * xhat)).
struct BnMask {
  // g = dz * act'(out): from the 1-byte mask of 4 channels when the forward wrote one, else from the activation output z
  static __device__ __forceinline__ f32x4 apply(f32x4 g, const unsigned char* mask, const float* z, long o, int act) {
    if (act == FS_ACT_NONE) return g;
    if (mask != nullptr) {
      const unsigned m = mask[o >> 2];
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] = ((m >> j) & 1u) ? g[j] : 0.f;
    } else {
      const f32x4 zz = *reinterpret_cast<const f32x4*>(z + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) g[j] *= fs_act_mask(zz[j], act);
    }
    return g;
  }
};

// column sums of one block's rows -> slab[blockIdx.x][C][2]; red = [rpi][C][2] floats of LDS: every row lane stores its partial sums and
// the columns are added over the row lanes in index order (no LDS atomics: their order is the waves' arrival order)
__device__ __forceinline__ void bn_slab_store(const RowWalk& w, f32x4 s, f32x4 sx, int C, int Ctot, float* __restrict__ slab, float* red) {
  if (w.active()) {
    float* p = red + ((long)w.r0 * C + 4 * w.col) * 2;
#pragma unroll
    for (int j = 0; j < 4; ++j) { p[2 * j] = s[j]; p[2 * j + 1] = sx[j]; }
  }
  __syncthreads();
  float* dst = slab + (long)blockIdx.x * Ctot * 2;
  for (int i = threadIdx.x; i < 2 * C; i += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < w.rpi; ++r) a += red[(long)r * 2 * C + i];
    dst[i] = a;
  }
}

__global__ __launch_bounds__(256)
void bn_bwd_partial_kernel(const float* __restrict__ dz, const float* __restrict__ z, const unsigned char* __restrict__ mask,
                           const float* __restrict__ y, const float* __restrict__ mean, const float* __restrict__ invstd, long M, int C,
                           int ld, int rows_per_block, int act, float* __restrict__ slab) {
  extern __shared__ float red[];
  RowWalk w(C);
  const int c = 4 * w.col;
  f32x4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  if (w.active()) {
    f32x4 mu, is;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
    const long rb = (long)blockIdx.x * rows_per_block;
    long re = rb + rows_per_block; if (re > M) re = M;
    for (long r = rb + w.r0; r < re; r += w.rpi) {
      const long o = r * ld + c;
      const f32x4 g = BnMask::apply(*reinterpret_cast<const f32x4*>(dz + o), mask, z, o, act);
      s += g; sx += g * ((*reinterpret_cast<const f32x4*>(y + o) - mu) * is);
    }
  }
  bn_slab_store(w, s, sx, C, ld, slab, red);
}

// out = a + b [+ c [+ d]]  (the gradient sum at a tensor with several consumers; ops.FanOut) AND, because that tensor is the
// output z of a conv + BatchNorm + activation layer, the layer's BatchNorm-backward column sums of the freshly formed gradient:
// the sum is in registers here, so the BatchNorm backward's own reduction pass over (dz, y) -- two of its five tensor passes -- is
// replaced by one extra read of y in this kernel.  Same row/column walk and slab layout as bn_bwd_partial_kernel.
__global__ __launch_bounds__(256)
void add_n_bnsum_kernel(const float* __restrict__ a, const float* __restrict__ b, const float* __restrict__ c_, const float* __restrict__ d,
                        float* __restrict__ out, const unsigned char* __restrict__ mask, const float* __restrict__ y,
                        const float* __restrict__ mean, const float* __restrict__ invstd, long M, int C, int ld, int rows_per_block, int act,
                        float* __restrict__ slab) {
  extern __shared__ float red[];
  RowWalk w(C);
  const int c = 4 * w.col;
  f32x4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  if (w.active()) {
    f32x4 mu, is;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
    const long rb = (long)blockIdx.x * rows_per_block;
    long re = rb + rows_per_block; if (re > M) re = M;
    for (long r = rb + w.r0; r < re; r += w.rpi) {
      const long o = r * ld + c;
      f32x4 v = *reinterpret_cast<const f32x4*>(a + o) + *reinterpret_cast<const f32x4*>(b + o);
      const f32x4 yy = *reinterpret_cast<const f32x4*>(y + o);
      if (c_ != nullptr) v += *reinterpret_cast<const f32x4*>(c_ + o);
      if (d != nullptr) v += *reinterpret_cast<const f32x4*>(d + o);
      *reinterpret_cast<f32x4*>(out + o) = v;
      const f32x4 g = BnMask::apply(v, mask, nullptr, o, mask != nullptr ? act : FS_ACT_NONE);
      s += g; sx += g * ((yy - mu) * is);
    }
  }
  bn_slab_store(w, s, sx, C, ld, slab, red);
}

// one block per channel: slab column -> S, SX (double tree) -> dbeta, dgamma and the apply coefficients coef[4][Ctot] = ga, d, mean, bb
__global__ __launch_bounds__(256) void bn_bwd_finalize_kernel(const float* __restrict__ slab, int nslab, const float* __restrict__ gamma,
                                                              const float* __restrict__ mean, const float* __restrict__ invstd, long M,
                                                              int C, int training, float* __restrict__ coef, float* __restrict__ dgamma,
                                                              float* __restrict__ dbeta, int accumulate) {
  __shared__ double red[16];
  const int c = blockIdx.x;
  double s = 0.0, sx = 0.0;
  for (int i = threadIdx.x; i < nslab; i += blockDim.x) {
    const float2 v = *reinterpret_cast<const float2*>(slab + ((long)i * C + c) * 2);
    s += (double)v.x; sx += (double)v.y;
  }
  s = block_sum<double>(s, red);
  sx = block_sum<double>(sx, red);
  if (threadIdx.x != 0) return;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + (float)s;
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + (float)sx;
  const float is = invstd[c], ga = gamma[c] * is;
  const float mg = training ? (float)(s / (double)M) : 0.f, mgx = training ? (float)(sx / (double)M) : 0.f;
  coef[c] = ga;
  coef[C + c] = ga * mgx * is;
  coef[2 * C + c] = mean[c];
  coef[3 * C + c] = ga * mg;
}

__global__ __launch_bounds__(256)
void bn_bwd_apply_kernel(const float* __restrict__ dz, const float* __restrict__ z, const unsigned char* __restrict__ mask,
                         const float* __restrict__ y, const float* __restrict__ coef, long M, int C, int ld, int c0, int rows_per_block,
                         int act, float drop_scale, uint32_t drop_thresh, uint32_t drop_key, float* __restrict__ dy,
                         float* __restrict__ dres) {
  RowWalk w(C);
  if (!w.active()) return;
  const int c = 4 * w.col;
  const f32x4 ga = *reinterpret_cast<const f32x4*>(coef + c0 + c), dd = *reinterpret_cast<const f32x4*>(coef + ld + c0 + c),
              mu = *reinterpret_cast<const f32x4*>(coef + 2 * ld + c0 + c), bb = *reinterpret_cast<const f32x4*>(coef + 3 * ld + c0 + c);
  // workgroups are dispatched in blockIdx order: the first ones take the LAST rows -- the rows the pass before read or wrote last and
  // the 256 MB Infinity Cache most likely still holds (dz + y of a 64-channel 80x80 layer are 210 MB)
  const long rb = (long)(gridDim.x - 1 - blockIdx.x) * rows_per_block;
  long re = rb + rows_per_block; if (re > M) re = M;
  auto finish = [&](long o, f32x4 g, const f32x4 yy) {
    if (dres != nullptr) *reinterpret_cast<f32x4*>(dres + o) = g;
    f32x4 v = ga * g - dd * (yy - mu) - bb;
    if (drop_thresh != 0u) {
      const uint32_t e = (uint32_t)(o + c0);                       // a multiple of 4 (channel quad of an NHWC row)
      const uint32_t keep = fs_dropout_keep4(e, drop_key, drop_thresh);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ((keep >> j) & 1u) ? v[j] * drop_scale : 0.f;
    }
    *reinterpret_cast<f32x4*>(dy + o) = v;
  };
  // one row per trip: 48 VGPRs keep eight of these waves per SIMD (as many bytes in flight as two rows at six) and let a workgroup
  // fit beside two 230-VGPR convolution waves
  for (long r = rb + w.r0; r < re; r += w.rpi) {
    const long o = r * ld + c;
    finish(o, BnMask::apply(*reinterpret_cast<const f32x4*>(dz + o), mask, z, o, act), *reinterpret_cast<const f32x4*>(y + o));
  }
}

// ---- bilinear source coordinates (align_corners=False, ATen upsample_bilinear2d) ------------
struct Lerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ Lerp lerp_coord(int d, int in, int out) {
  Lerp L;
  if (in == out) { L.i0 = d; L.i1 = d; L.l0 = 1.f; L.l1 = 0.f; return L; }
  const float scale = (float)in / (float)out;
  float s = scale * ((float)d + 0.5f) - 0.5f;
  if (s < 0.f) s = 0.f;
  L.i0 = (int)s;
  L.i1 = L.i0 + (L.i0 < in - 1 ? 1 : 0);
  L.l1 = s - (float)L.i0;
  L.l0 = 1.f - L.l1;
  return L;
}

__device__ __forceinline__ f32x4 bilerp4(const float* __restrict__ src, int b, int th, int tw, int C, int c,
                                         const Lerp& Ly, const Lerp& Lx) {
  const float* base = src + (long)b * th * tw * C + c;
  const f32x4 v00 = *reinterpret_cast<const f32x4*>(base + ((long)Ly.i0 * tw + Lx.i0) * C);
  const f32x4 v01 = *reinterpret_cast<const f32x4*>(base + ((long)Ly.i0 * tw + Lx.i1) * C);
  const f32x4 v10 = *reinterpret_cast<const f32x4*>(base + ((long)Ly.i1 * tw + Lx.i0) * C);
  const f32x4 v11 = *reinterpret_cast<const f32x4*>(base + ((long)Ly.i1 * tw + Lx.i1) * C);
  return Ly.l0 * (Lx.l0 * v00 + Lx.l1 * v01) + Ly.l1 * (Lx.l0 * v10 + Lx.l1 * v11);
}

// ---- HRNet fuse: out = relu(t0 + t1 + t2 + t3), lower-resolution terms bilinearly up-sampled ---
struct FuseArgs {
  const float* t[4];
  int th[4], tw[4];
  int nterms;
  float* out;
  int B, Ho, Wo, C, relu;
};
__global__ __launch_bounds__(256) void hr_fuse_fwd_kernel(FuseArgs a) {
  const int cw = a.C >> 2;
  const long total = (long)a.B * a.Ho * a.Wo * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % cw);
    const long pix = i / cw;
    const int ox = (int)(pix % a.Wo);
    const int oy = (int)((pix / a.Wo) % a.Ho);
    const int b = (int)(pix / ((long)a.Wo * a.Ho));
    f32x4 acc = {0, 0, 0, 0};
    for (int t = 0; t < a.nterms; ++t) {
      f32x4 v;
      if (a.th[t] == a.Ho && a.tw[t] == a.Wo) {
        v = *reinterpret_cast<const f32x4*>(a.t[t] + pix * a.C + c);
      } else {
        const Lerp Ly = lerp_coord(oy, a.th[t], a.Ho), Lx = lerp_coord(ox, a.tw[t], a.Wo);
        v = bilerp4(a.t[t], b, a.th[t], a.tw[t], a.C, c, Ly, Lx);
      }
      acc = (t == 0) ? v : acc + v;
    }
    if (a.relu) {
#pragma unroll
      for (int j = 0; j < 4; ++j) acc[j] = acc[j] < 0.f ? 0.f : acc[j];
    }
    *reinterpret_cast<f32x4*>(a.out + pix * a.C + c) = acc;
  }
}

// out = a + b [+ c [+ d]] (the gradient sum at a tensor with several consumers); out may alias a.  c, d nullable.
__global__ __launch_bounds__(256) void add_n_kernel(const float* a, const float* __restrict__ b, const float* __restrict__ c,
                                                    const float* __restrict__ d, float* out, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(a)[i] + reinterpret_cast<const f32x4*>(b)[i];
    if (c != nullptr) v += reinterpret_cast<const f32x4*>(c)[i];
    if (d != nullptr) v += reinterpret_cast<const f32x4*>(d)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

// g = dout * (out > 0)
__global__ __launch_bounds__(256) void relu_bwd_kernel(const float* __restrict__ dout, const float* __restrict__ out,
                                                       float* __restrict__ g, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 d = reinterpret_cast<const f32x4*>(dout)[i];
    const f32x4 o = reinterpret_cast<const f32x4*>(out)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) d[j] = o[j] > 0.f ? d[j] : 0.f;
    reinterpret_cast<f32x4*>(g)[i] = d;
  }
}

// g = dout * (out > 0) AND the BatchNorm-backward column sums of up to three conv + BatchNorm layers whose output gradient g IS (round 5: the
// same-resolution terms of an HRNet fuse row -- the last, activation-free ConvBn of every down-path -- all receive this one tensor): the
// gradient is in registers here, so each layer's own reduction pass over (g, y) becomes one extra read of its y.  Row / column walk and
// slab layout of bn_bwd_partial_kernel; sum g is the same for every term and is stored with each.
struct BnTerms { const float* y[3]; const float* mean[3]; const float* invstd[3]; float* slab[3]; int n; };
__global__ __launch_bounds__(256)
void relu_bwd_bnsum_kernel(const float* __restrict__ dout, const float* __restrict__ out, float* __restrict__ g, long M, int C,
                           int rows_per_block, BnTerms t) {
  extern __shared__ float red[];
  RowWalk w(C);
  const int c = 4 * w.col;
  f32x4 s = {0, 0, 0, 0}, sx[3] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
  if (w.active()) {
    f32x4 mu[3], is[3];
#pragma unroll
    for (int k = 0; k < 3; ++k)
#pragma unroll
      for (int j = 0; j < 4; ++j) { mu[k][j] = k < t.n ? t.mean[k][c + j] : 0.f; is[k][j] = k < t.n ? t.invstd[k][c + j] : 0.f; }
    const long rb = (long)blockIdx.x * rows_per_block;
    long re = rb + rows_per_block; if (re > M) re = M;
    for (long r = rb + w.r0; r < re; r += w.rpi) {
      const long o = r * C + c;
      f32x4 d = *reinterpret_cast<const f32x4*>(dout + o);
      const f32x4 z = *reinterpret_cast<const f32x4*>(out + o);
#pragma unroll
      for (int j = 0; j < 4; ++j) d[j] = z[j] > 0.f ? d[j] : 0.f;
      *reinterpret_cast<f32x4*>(g + o) = d;
      s += d;
#pragma unroll
      for (int k = 0; k < 3; ++k)
        if (k < t.n) sx[k] += d * ((*reinterpret_cast<const f32x4*>(t.y[k] + o) - mu[k]) * is[k]);
    }
  }
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    if (k < t.n) {                          // (workgroup-uniform)
      if (k > 0) __syncthreads();           // the previous term's readers of `red` are done
      bn_slab_store(w, s, sx[k], C, C, t.slab[k], red);
    }
  }
}

// ---- bilinear up-sample into a channel slice of a wider NHWC buffer (final 960-ch concat) ------
__global__ __launch_bounds__(256) void upsample_slice_fwd_kernel(const float* __restrict__ src, int B, int th, int tw, int C,
                                                                 float* __restrict__ dst, int Ho, int Wo, int Cdst, int coff) {
  const int cw = C >> 2;
  const long total = (long)B * Ho * Wo * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % cw);
    const long pix = i / cw;
    const int ox = (int)(pix % Wo);
    const int oy = (int)((pix / Wo) % Ho);
    const int b = (int)(pix / ((long)Wo * Ho));
    const Lerp Ly = lerp_coord(oy, th, Ho), Lx = lerp_coord(ox, tw, Wo);
    *reinterpret_cast<f32x4*>(dst + pix * Cdst + coff + c) = bilerp4(src, b, th, tw, C, c, Ly, Lx);
  }
}

// transpose of the above as a gather: dsrc[b,qy,qx,c] = sum over the <= (2f)^2 output pixels whose
// 2x2 footprint touches (qy,qx).  g is read from a channel slice (stride Cg, offset coff).
__global__ __launch_bounds__(256) void upsample_slice_bwd_kernel(const float* __restrict__ g, int B, int Ho, int Wo, int Cg,
                                                                 int coff, float* __restrict__ dsrc, int th, int tw, int C) {
  const int cw = C >> 2;
  const long total = (long)B * th * tw * cw;
  const int fy = Ho / th, fx = Wo / tw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % cw);
    const long q = i / cw;
    const int qx = (int)(q % tw);
    const int qy = (int)((q / tw) % th);
    const int b = (int)(q / ((long)tw * th));
    f32x4 acc = {0, 0, 0, 0};
    if (fy == 1 && fx == 1) {
      acc = *reinterpret_cast<const f32x4*>(g + (((long)b * Ho + qy) * Wo + qx) * Cg + coff + c);
    } else {
      int y_lo = fy * qy - fy / 2, y_hi = fy * qy + (3 * fy) / 2 - 1;
      int x_lo = fx * qx - fx / 2, x_hi = fx * qx + (3 * fx) / 2 - 1;
      if (y_lo < 0) y_lo = 0;
      if (x_lo < 0) x_lo = 0;
      if (y_hi > Ho - 1 || qy == th - 1) y_hi = Ho - 1;
      if (x_hi > Wo - 1 || qx == tw - 1) x_hi = Wo - 1;
      for (int oy = y_lo; oy <= y_hi; ++oy) {
        const Lerp Ly = lerp_coord(oy, th, Ho);
        const float wy = (Ly.i0 == qy ? Ly.l0 : 0.f) + (Ly.i1 == qy ? Ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = x_lo; ox <= x_hi; ++ox) {
          const Lerp Lx = lerp_coord(ox, tw, Wo);
          const float wx = (Lx.i0 == qx ? Lx.l0 : 0.f) + (Lx.i1 == qx ? Lx.l1 : 0.f);
          if (wx == 0.f) continue;
          acc += (wy * wx) * *reinterpret_cast<const f32x4*>(g + (((long)b * Ho + oy) * Wo + ox) * Cg + coff + c);
        }
      }
    }
    *reinterpret_cast<f32x4*>(dsrc + q * C + c) = acc;
  }
}

// the same gather with the row / column walk of the BatchNorm kernels, AND the BatchNorm-backward column sums of the layer whose output
// gradient dsrc is (round 5: the 1x1 ConvBn of an HRNet fuse up-path, which is activation-free): sum d, sum d * xhat into slab[block][C][2]
__global__ __launch_bounds__(256)
void upsample_slice_bwd_bnsum_kernel(const float* __restrict__ g, int Ho, int Wo, int Cg, int coff, float* __restrict__ dsrc, int th, int tw,
                                     int C, long M, int rows_per_block, const float* __restrict__ y, const float* __restrict__ mean,
                                     const float* __restrict__ invstd, float* __restrict__ slab) {
  extern __shared__ float red[];
  RowWalk w(C);
  const int c = 4 * w.col;
  const int fy = Ho / th, fx = Wo / tw;
  f32x4 s = {0, 0, 0, 0}, sx = {0, 0, 0, 0};
  if (w.active()) {
    f32x4 mu, is;
#pragma unroll
    for (int j = 0; j < 4; ++j) { mu[j] = mean[c + j]; is[j] = invstd[c + j]; }
    const long rb = (long)blockIdx.x * rows_per_block;
    long re = rb + rows_per_block; if (re > M) re = M;
    for (long q = rb + w.r0; q < re; q += w.rpi) {
      const int qx = (int)(q % tw);
      const int qy = (int)((q / tw) % th);
      const int b = (int)(q / ((long)tw * th));
      f32x4 acc = {0, 0, 0, 0};
      int y_lo = fy * qy - fy / 2, y_hi = fy * qy + (3 * fy) / 2 - 1;
      int x_lo = fx * qx - fx / 2, x_hi = fx * qx + (3 * fx) / 2 - 1;
      if (y_lo < 0) y_lo = 0;
      if (x_lo < 0) x_lo = 0;
      if (y_hi > Ho - 1 || qy == th - 1) y_hi = Ho - 1;
      if (x_hi > Wo - 1 || qx == tw - 1) x_hi = Wo - 1;
      for (int oy = y_lo; oy <= y_hi; ++oy) {
        const Lerp Ly = lerp_coord(oy, th, Ho);
        const float wy = (Ly.i0 == qy ? Ly.l0 : 0.f) + (Ly.i1 == qy ? Ly.l1 : 0.f);
        if (wy == 0.f) continue;
        for (int ox = x_lo; ox <= x_hi; ++ox) {
          const Lerp Lx = lerp_coord(ox, tw, Wo);
          const float wx = (Lx.i0 == qx ? Lx.l0 : 0.f) + (Lx.i1 == qx ? Lx.l1 : 0.f);
          if (wx == 0.f) continue;
          acc += (wy * wx) * *reinterpret_cast<const f32x4*>(g + (((long)b * Ho + oy) * Wo + ox) * Cg + coff + c);
        }
      }
      const long o = q * C + c;
      *reinterpret_cast<f32x4*>(dsrc + o) = acc;
      s += acc; sx += acc * ((*reinterpret_cast<const f32x4*>(y + o) - mu) * is);
    }
  }
  bn_slab_store(w, s, sx, C, C, slab, red);
}

// ---- column sums (bias gradients): per-row-block partials into part[block][C], summed in block order by fs_slab_reduce ----------
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, long M, int C, int rows_per_block,
                                                     float* __restrict__ part) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const long rb = (long)blockIdx.y * rows_per_block;
  long re = rb + rows_per_block; if (re > M) re = M;
  float s = 0.f;
  for (long r = rb; r < re; ++r) s += x[r * C + c];
  part[(long)blockIdx.y * C + c] = s;
}

// the same for C % 4 == 0, C <= 1024: 16-byte loads over the RowWalk layout (all 256 threads busy for any C, four rows in flight),
// per-workgroup partials through LDS, one store per column and workgroup
__global__ __launch_bounds__(256) void colsum4_kernel(const float* __restrict__ x, long M, int C, int rows_per_block, float* __restrict__ part) {
  extern __shared__ float csm[];   // [rpi][C]
  RowWalk w(C);
  const int c = 4 * w.col;
  f32x4 s = {0.f, 0.f, 0.f, 0.f};
  if (w.active()) {
    const long rb = (long)blockIdx.x * rows_per_block;
    long re = rb + rows_per_block; if (re > M) re = M;
    long r = rb + w.r0;
    const long step = w.rpi;
    for (; r + 3 * step < re; r += 4 * step) {
      const f32x4 a0 = *reinterpret_cast<const f32x4*>(x + r * C + c);
      const f32x4 a1 = *reinterpret_cast<const f32x4*>(x + (r + step) * C + c);
      const f32x4 a2 = *reinterpret_cast<const f32x4*>(x + (r + 2 * step) * C + c);
      const f32x4 a3 = *reinterpret_cast<const f32x4*>(x + (r + 3 * step) * C + c);
      s += a0; s += a1; s += a2; s += a3;
    }
    for (; r < re; r += step) s += *reinterpret_cast<const f32x4*>(x + r * C + c);
    *reinterpret_cast<f32x4*>(&csm[w.r0 * C + c]) = s;
  }
  __syncthreads();
  for (int cc = threadIdx.x; cc < C; cc += blockDim.x) {
    float a = 0.f;
    for (int r = 0; r < w.rpi; ++r) a += csm[r * C + cc];
    part[(long)blockIdx.x * C + cc] = a;
  }
}

// ---- global average pool over HW (AvgPool2d((10,10)) on a 10x10 map) ---------------------------
__global__ void avgpool_fwd_kernel(const float* __restrict__ x, int B, int HW, int C, float* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * C) return;
  const int b = i / C, c = i - b * C;
  float s = 0.f;
  for (int p = 0; p < HW; ++p) s += x[((long)b * HW + p) * C + c];
  out[i] = s / (float)HW;
}
__global__ void avgpool_bwd_kernel(const float* __restrict__ dout, int B, int HW, int C, float* __restrict__ dx) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW * C) return;
  const int c = (int)(i % C);
  const int b = (int)(i / ((long)HW * C));
  dx[i] = dout[b * C + c] / (float)HW;
}

// ---- max pooling (NHWC), argmax kept as the flat input pixel index for the backward ------------
__global__ __launch_bounds__(256) void maxpool_fwd_kernel(const float* __restrict__ x, float* __restrict__ out,
                                                          int* __restrict__ arg, int B, int H, int W, int C, int Ho, int Wo,
                                                          int k, int stride, int pad) {
  const long total = (long)B * Ho * Wo * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int ox = (int)(pix % Wo), oy = (int)((pix / Wo) % Ho), b = (int)(pix / ((long)Wo * Ho));
    float best = -INFINITY; int bi = -1;
    for (int r = 0; r < k; ++r) {
      const int iy = oy * stride - pad + r;
      if (iy < 0 || iy >= H) continue;
      for (int q = 0; q < k; ++q) {
        const int ix = ox * stride - pad + q;
        if (ix < 0 || ix >= W) continue;
        const float v = x[(((long)b * H + iy) * W + ix) * C + c];
        if (v > best || bi < 0 || v != v) { best = v; bi = iy * W + ix; }
      }
    }
    out[i] = best; arg[i] = bi;
  }
}
// gather form of the scatter: every input pixel looks at the (<= ceil(k/stride)^2) windows covering it
__global__ __launch_bounds__(256) void maxpool_bwd_kernel(const float* __restrict__ dout, const int* __restrict__ arg,
                                                          float* __restrict__ dx, int B, int H, int W, int C, int Ho, int Wo,
                                                          int k, int stride, int pad) {
  const long total = (long)B * H * W * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    const long pix = i / C;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    const int me = iy * W + ix;
    int oy_lo = (iy + pad - k + stride) / stride; if (iy + pad - k + 1 < 0) oy_lo = 0;
    int ox_lo = (ix + pad - k + stride) / stride; if (ix + pad - k + 1 < 0) ox_lo = 0;
    int oy_hi = (iy + pad) / stride; if (oy_hi > Ho - 1) oy_hi = Ho - 1;
    int ox_hi = (ix + pad) / stride; if (ox_hi > Wo - 1) ox_hi = Wo - 1;
    float acc = 0.f;
    for (int oy = oy_lo; oy <= oy_hi; ++oy)
      for (int ox = ox_lo; ox <= ox_hi; ++ox) {
        const long o = (((long)b * Ho + oy) * Wo + ox) * C + c;
        if (arg[o] == me) acc += dout[o];
      }
    dx[i] = acc;
  }
}

// ---- stand-alone Dropout (ASPP projection): out = keep ? x/(1-p) : 0, same hash as the conv epilogue ----
__global__ __launch_bounds__(256) void dropout_kernel(const float* __restrict__ x, float* __restrict__ out, long n,
                                                      float scale, uint32_t thresh, uint32_t key) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = fs_dropout_keep((uint32_t)i, key, thresh) ? x[i] * scale : 0.f;
}

int rows_per_block_for(long M, int C) {
  const int cw = C / 4;
  int rpi = 256 / cw; if (rpi < 1) rpi = 1;
  long rpb = (M + 1023) / 1024;
  rpb = ((rpb + rpi - 1) / rpi) * rpi;
  if (rpb < rpi) rpb = rpi;
  return (int)rpb;
}
size_t stats_smem(int C) {
  const int cw = C / 4;
  int rpi = 256 / cw; if (rpi < 1) rpi = 1;
  return (size_t)rpi * C * 2 * sizeof(double);
}

__global__ __launch_bounds__(256) void slab_reduce_kernel(const float* __restrict__ part, int nslab, long n, float* __restrict__ out, int accumulate) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float s = accumulate ? out[i] : 0.f;
  for (int k = 0; k < nslab; ++k) s += part[(long)k * n + i];
  out[i] = s;
}
// few columns, many slabs (bias / 1x1-head gradients from ~1000 workgroups): one wave per column, lane j adds slabs j, j + 64, ... in
// order and the 64 lane sums meet in a fixed shuffle tree -- the order depends on (nslab) only, never on timing
__global__ __launch_bounds__(256) void slab_reduce_wave_kernel(const float* __restrict__ part, int nslab, long n, float* __restrict__ out, int accumulate) {
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
  for (int k = lane; k < nslab; k += 64) s += part[(long)k * n + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = accumulate ? out[i] + s : s;
}

// many slabs of a mid-sized tensor (512 partial dW tiles of a 64 x 64 x 9 layer: 36 864 elements each): one thread per element walking 512
// slabs leaves the chip to 144 workgroups of serial loads.  Stage 1 lets G workgroup rows (blockIdx.y = g) each add the slabs k = g, g + G, ...
// of their elements in ascending k and leaves the sum IN slab g (the only reader of slab g's element i is the thread that overwrites
// it); stage 2 is the plain kernel over the first G slabs.  The order is fixed by (nslab, G): still independent of timing.
__global__ __launch_bounds__(256) void slab_reduce_stage1_kernel(float* __restrict__ part, int nslab, long n, int G) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  const int g = blockIdx.y;
  if (i >= n) return;
  float s = 0.f;
  for (int k = g; k < nslab; k += G) s += part[(long)k * n + i];
  part[(long)g * n + i] = s;
}

// two such sums of equal shape in one launch (blockIdx.y picks the pair member): LayerNorm's dgamma / dbeta, a head's dw / db
__global__ __launch_bounds__(256) void slab_reduce_pair_kernel(const float* __restrict__ part0, const float* __restrict__ part1, int nslab, long n,
                                                               float* __restrict__ out0, float* __restrict__ out1, int accumulate) {
  const float* part = blockIdx.y ? part1 : part0;
  float* out = blockIdx.y ? out1 : out0;
  const long i = ((long)blockIdx.x * 256 + threadIdx.x) >> 6;
  const int lane = threadIdx.x & 63;
  if (i >= n) return;
  float s = 0.f;
  for (int k = lane; k < nslab; k += 64) s += part[(long)k * n + i];
  s = wave_sum(s);
  if (lane == 0) out[i] = accumulate ? out[i] + s : s;
}

}  // namespace

int fs_slab_reduce_pair(const float* part0, const float* part1, int nslab, long n, float* out0, float* out1, int accumulate, hipStream_t stream) {
  if (!part0 || !part1 || !out0 || !out1 || nslab < 0 || n <= 0) return FS_ERR_ARG;
  hipLaunchKernelGGL(slab_reduce_pair_kernel, dim3((unsigned)((n * 64 + 255) / 256), 2), dim3(256), 0, stream, part0, part1, nslab, n, out0, out1, accumulate);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_slab_reduce_inplace(float* part, int nslab, long n, float* out, int accumulate, hipStream_t stream) {
  if (part == nullptr || out == nullptr || nslab < 0 || n <= 0) return FS_ERR_ARG;
  constexpr int G = 8;
  if (nslab >= 4 * G && n > 4096 && n * (long)nslab >= (1L << 22)) {
    hipLaunchKernelGGL(slab_reduce_stage1_kernel, dim3((unsigned)((n + 255) / 256), G), dim3(256), 0, stream, part, nslab, n, G);
    FS_LAUNCH_CHECK();
    nslab = G;
  }
  return fs_slab_reduce(part, nslab, n, out, accumulate, stream);
}

// common.h: the ordered second stage of every cross-workgroup sum
int fs_slab_reduce(const float* part, int nslab, long n, float* out, int accumulate, hipStream_t stream) {
  if (part == nullptr || out == nullptr || nslab < 0 || n <= 0) return FS_ERR_ARG;
  if (n <= 4096 && nslab > 64)
    hipLaunchKernelGGL(slab_reduce_wave_kernel, dim3((unsigned)((n * 64 + 255) / 256)), dim3(256), 0, stream, part, nslab, n, out, accumulate);
  else
    hipLaunchKernelGGL(slab_reduce_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, stream, part, nslab, n, out, accumulate);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

extern "C" {

// include/fovealseg.h: fs_stream_wait -- everything enqueued on `waiter` after this call runs after everything enqueued on `signaller`
// before it (hipEventRecord + hipStreamWaitEvent on an event from a per-thread ring; a wait captures the record that precedes it, so
// an event can be recorded again as soon as its wait has been enqueued).  One host call instead of torch's Stream / Event round trip.
int fs_stream_wait(hipStream_t waiter, hipStream_t signaller) {
  constexpr int RING = 64;
  static thread_local hipEvent_t ring[RING];
  static thread_local int ring_dev[RING];
  static thread_local int ring_n = 0, next = 0;
  if (waiter == signaller) return FS_OK;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
  if (ring_n < RING) {
    if (hipEventCreateWithFlags(&ring[ring_n], hipEventDisableTiming) != hipSuccess) return FS_ERR_ARG;
    ring_dev[ring_n] = dev;
    next = ring_n++;
  } else {
    next = (next + 1) % RING;
    if (ring_dev[next] != dev) {           // the thread moved to another device: the slot's event belongs to the old one
      (void)hipEventDestroy(ring[next]);
      if (hipEventCreateWithFlags(&ring[next], hipEventDisableTiming) != hipSuccess) return FS_ERR_ARG;
      ring_dev[next] = dev;
    }
  }
  hipError_t e = hipEventRecord(ring[next], signaller);
  if (e != hipSuccess) return (int)e;
  e = hipStreamWaitEvent(waiter, ring[next], 0);
  return e == hipSuccess ? FS_OK : (int)e;
}

// Training-mode statistics of y (M rows, C channels): mean/invstd out, running stats updated in
// place with `momentum` (unbiased variance), `sums` = 2*C doubles of scratch.
// row blocks of fs_bn_stats (the widest count over its channel windows) = records of 2*C doubles in its scratch
static int bn_stats_blocks(long M, int C) {
  int nblk = 1;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int n = cdiv(M, rows_per_block_for(M, Cc));
    if (n > nblk) nblk = n;
  }
  return nblk;
}
long fs_bn_stats_scratch_doubles(long M, int C) { return (M > 0 && C > 0) ? (long)bn_stats_blocks(M, C) * 2 * C : 0; }

int fs_bn_stats(const float* y, long M, int C, float momentum, float eps, float* running_mean, float* running_var,
                float* mean, float* invstd, double* sums, hipStream_t stream) {
  FS_REQUIRE(y && mean && invstd && sums && M > 0 && C > 0 && C % 4 == 0);
  const int nblk = bn_stats_blocks(M, C);
  hipError_t e = hipMemsetAsync(sums, 0, (size_t)nblk * 2 * C * sizeof(double), stream);       // windows with fewer row blocks leave records untouched
  if (e != hipSuccess) return (int)e;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int rpb = rows_per_block_for(M, Cc);
    hipLaunchKernelGGL(bn_stats_kernel, dim3(cdiv(M, rpb)), dim3(256), stats_smem(Cc), stream, y + c0, M, Cc, C, C, rpb, sums + c0);
    FS_LAUNCH_CHECK();
  }
  hipLaunchKernelGGL(bn_finalize_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, sums, nblk, M, C, momentum, eps,
                     running_mean, running_var, mean, invstd);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// Finalise training-mode statistics from the conv epilogue's partial slab ([nwg][C][2] floats).
int fs_bn_finalize_slab(const float* slab, int nwg, long M, int C, float momentum, float eps, float* running_mean,
                        float* running_var, float* mean, float* invstd, hipStream_t stream) {
  FS_REQUIRE(slab && mean && invstd && nwg > 0 && M > 0 && C > 0);
  hipLaunchKernelGGL(bn_finalize_slab_kernel, dim3(C), dim3(256), 0, stream, slab, nwg, M, C, momentum, eps, running_mean,
                     running_var, mean, invstd);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_bn_eval_prepare(const float* running_mean, const float* running_var, int C, float eps, float* mean, float* invstd,
                       hipStream_t stream) {
  FS_REQUIRE(running_mean && running_var && mean && invstd && C > 0);
  hipLaunchKernelGGL(bn_eval_prepare_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var, C, eps,
                     mean, invstd);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_bn_eval_affine(const float* running_mean, const float* running_var, const float* gamma, const float* beta, int C, float eps,
                      float* scale, float* shift, hipStream_t stream) {
  FS_REQUIRE(running_mean && running_var && gamma && beta && scale && shift && C > 0);
  hipLaunchKernelGGL(bn_eval_affine_kernel, dim3(cdiv(C, 256)), dim3(256), 0, stream, running_mean, running_var, gamma, beta, C, eps, scale,
                     shift);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_bn_act_fwd(const float* y, const float* mean, const float* invstd, const float* gamma, const float* beta,
                  const float* res, float* out, unsigned char* mask, long M, int C, int act, hipStream_t stream) {
  FS_REQUIRE(y && mean && invstd && gamma && beta && out && M > 0 && C > 0 && C % 4 == 0);
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int rpb = rows_per_block_for(M, Cc);
    hipLaunchKernelGGL(bn_act_fwd_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, stream, y + c0, mean + c0, invstd + c0, gamma + c0,
                       beta + c0, res ? res + c0 : nullptr, out + c0, mask ? mask + c0 / 4 : nullptr, M, Cc, C, rpb, act);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

// ---- Backward of out = act(bn(y) + res): partial sums -> finalize -> apply (kernels above) ------------------------------------------
// Number of slab rows ([.][C][2] floats each) fs_bn_bwd_partial / fs_add_n_bnsum write for an (M, C) activation.
int fs_bn_bwd_slabs(long M, int C) {
  if (M <= 0 || C <= 0 || C % 4) return 0;
  int n = 0;
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int k = cdiv(M, rows_per_block_for(M, Cc));
    n = k > n ? k : n;
  }
  return n;
}

// LDS of bn_slab_store: [rpi][C][2] floats, rpi = row lanes of the RowWalk layout (<= 8.2 KB)
static size_t slab_lds_bytes(int C) {
  int rpi = 256 / (C / 4); if (rpi < 1) rpi = 1;
  return (size_t)rpi * C * 2 * sizeof(float);
}

int fs_bn_bwd_partial(const float* dz, const float* z, const unsigned char* mask, const float* y, const float* mean, const float* invstd,
                      long M, int C, int act, float* slab, hipStream_t stream) {
  FS_REQUIRE(dz && y && mean && invstd && slab && M > 0 && C > 0 && C % 4 == 0);
  FS_REQUIRE(act == FS_ACT_NONE || z != nullptr || mask != nullptr);
  const int nslab = fs_bn_bwd_slabs(M, C);
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int rpb = rows_per_block_for(M, Cc);
    // every launch writes all nslab rows of its channel window (blocks past the last row write zeros)
    hipLaunchKernelGGL(bn_bwd_partial_kernel, dim3(nslab), dim3(256), slab_lds_bytes(Cc), stream, dz + c0, z ? z + c0 : nullptr,
                       mask ? mask + c0 / 4 : nullptr, y + c0, mean + c0, invstd + c0, M, Cc, C, rpb, act, slab + 2 * c0);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

int fs_add_n_bnsum(const float* a, const float* b, const float* c, const float* d, float* out, const unsigned char* mask, const float* y,
                   const float* mean, const float* invstd, long M, int C, int act, float* slab, hipStream_t stream) {
  FS_REQUIRE(a && b && out && y && mean && invstd && slab && M > 0 && C > 0 && C % 4 == 0 && (d == nullptr || c != nullptr));
  FS_REQUIRE(act == FS_ACT_NONE || mask != nullptr);
  const int nslab = fs_bn_bwd_slabs(M, C);
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int rpb = rows_per_block_for(M, Cc);
    hipLaunchKernelGGL(add_n_bnsum_kernel, dim3(nslab), dim3(256), slab_lds_bytes(Cc), stream, a + c0, b + c0, c ? c + c0 : nullptr,
                       d ? d + c0 : nullptr, out + c0, mask ? mask + c0 / 4 : nullptr, y + c0, mean + c0, invstd + c0, M, Cc, C, rpb, act,
                       slab + 2 * c0);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

int fs_bn_bwd_finalize(const float* slab, int nslab, const float* gamma, const float* mean, const float* invstd, long M, int C,
                       int training, float* coef, float* dgamma, float* dbeta, int accumulate_affine, hipStream_t stream) {
  FS_REQUIRE(slab && nslab > 0 && gamma && mean && invstd && coef && dgamma && dbeta && M > 0 && C > 0);
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3(C), dim3(256), 0, stream, slab, nslab, gamma, mean, invstd, M, C, training, coef, dgamma,
                     dbeta, accumulate_affine);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_bn_bwd_apply(const float* dz, const float* z, const unsigned char* mask, const float* y, const float* coef, long M, int C, int act,
                    float drop_p, uint32_t drop_key, float* dy, float* dres, hipStream_t stream) {
  FS_REQUIRE(dz && y && coef && dy && M > 0 && C > 0 && C % 4 == 0);
  FS_REQUIRE(act == FS_ACT_NONE || z != nullptr || mask != nullptr);
  float scale = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { scale = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  for (int c0 = 0; c0 < C; c0 += 1024) {
    const int Cc = C - c0 < 1024 ? C - c0 : 1024;
    const int rpb = rows_per_block_for(M, Cc);
    hipLaunchKernelGGL(bn_bwd_apply_kernel, dim3(cdiv(M, rpb)), dim3(256), 0, stream, dz + c0, z ? z + c0 : nullptr,
                       mask ? mask + c0 / 4 : nullptr, y + c0, coef, M, Cc, C, c0, rpb, act, scale, thresh, drop_key, dy + c0,
                       dres ? dres + c0 : nullptr);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

int fs_hr_fuse_fwd(const float* const* terms, const int* th, const int* tw, int nterms, float* out, int B, int Ho, int Wo,
                   int C, int relu, hipStream_t stream) {
  FS_REQUIRE(terms && th && tw && out && nterms >= 1 && nterms <= 4 && C % 4 == 0);
  FuseArgs a;
  for (int t = 0; t < 4; ++t) { a.t[t] = nullptr; a.th[t] = 0; a.tw[t] = 0; }
  for (int t = 0; t < nterms; ++t) {
    FS_REQUIRE(terms[t] && th[t] > 0 && tw[t] > 0 && Ho % th[t] == 0 && Wo % tw[t] == 0);
    a.t[t] = terms[t]; a.th[t] = th[t]; a.tw[t] = tw[t];
  }
  a.nterms = nterms; a.out = out; a.B = B; a.Ho = Ho; a.Wo = Wo; a.C = C; a.relu = relu;
  const long total = (long)B * Ho * Wo * (C / 4);
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(hr_fuse_fwd_kernel, dim3(blocks), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_add_n
int fs_add_n(const float* a, const float* b, const float* c, const float* d, float* out, long n, hipStream_t stream) {
  FS_REQUIRE(a && b && out && n > 0 && n % 4 == 0 && (d == nullptr || c != nullptr));
  int blocks = cdiv(n / 4, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(add_n_kernel, dim3(blocks), dim3(256), 0, stream, a, b, c, d, out, n / 4);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_relu_bwd(const float* dout, const float* out, float* g, long n, hipStream_t stream) {
  FS_REQUIRE(dout && out && g && n > 0 && n % 4 == 0);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(relu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dout, out, g, n / 4);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_relu_bwd_bnsum -- g = dout * (out > 0) over (M, C) plus, for nterm <= 3 layers, slab_k[fs_bn_bwd_slabs(M, C)][C][2]
int fs_relu_bwd_bnsum(const float* dout, const float* out, float* g, long M, int C, int nterm, const float* const* y, const float* const* mean,
                      const float* const* invstd, float* const* slab, hipStream_t stream) {
  FS_REQUIRE(dout && out && g && M > 0 && C > 0 && C % 4 == 0 && C <= 1024 && nterm >= 1 && nterm <= 3 && y && mean && invstd && slab);
  BnTerms t;
  t.n = nterm;
  for (int k = 0; k < 3; ++k) {
    const bool on = k < nterm;
    FS_REQUIRE(!on || (y[k] && mean[k] && invstd[k] && slab[k]));
    t.y[k] = on ? y[k] : nullptr; t.mean[k] = on ? mean[k] : nullptr; t.invstd[k] = on ? invstd[k] : nullptr; t.slab[k] = on ? slab[k] : nullptr;
  }
  const int rpb = rows_per_block_for(M, C);
  hipLaunchKernelGGL(relu_bwd_bnsum_kernel, dim3(fs_bn_bwd_slabs(M, C)), dim3(256), slab_lds_bytes(C), stream, dout, out, g, M, C, rpb, t);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_upsample_slice_fwd(const float* src, int B, int th, int tw, int C, float* dst, int Ho, int Wo, int Cdst, int coff,
                          hipStream_t stream) {
  FS_REQUIRE(src && dst && C % 4 == 0 && Cdst % 4 == 0 && coff % 4 == 0 && coff + C <= Cdst && Ho % th == 0 && Wo % tw == 0);
  const long total = (long)B * Ho * Wo * (C / 4);
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(upsample_slice_fwd_kernel, dim3(blocks), dim3(256), 0, stream, src, B, th, tw, C, dst, Ho, Wo, Cdst, coff);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_upsample_slice_bwd(const float* g, int B, int Ho, int Wo, int Cg, int coff, float* dsrc, int th, int tw, int C,
                          hipStream_t stream) {
  FS_REQUIRE(g && dsrc && C % 4 == 0 && Cg % 4 == 0 && coff % 4 == 0 && coff + C <= Cg && Ho % th == 0 && Wo % tw == 0);
  FS_REQUIRE((Ho / th == 1 || (Ho / th) % 2 == 0) && (Wo / tw == 1 || (Wo / tw) % 2 == 0));
  const long total = (long)B * th * tw * (C / 4);
  int blocks = cdiv(total, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(upsample_slice_bwd_kernel, dim3(blocks), dim3(256), 0, stream, g, B, Ho, Wo, Cg, coff, dsrc, th, tw, C);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_upsample_slice_bwd_bnsum -- fs_upsample_slice_bwd (true up-sampling factors only) plus slab[fs_bn_bwd_slabs(B*th*tw, C)][C][2]
int fs_upsample_slice_bwd_bnsum(const float* g, int B, int Ho, int Wo, int Cg, int coff, float* dsrc, int th, int tw, int C, const float* y,
                                const float* mean, const float* invstd, float* slab, hipStream_t stream) {
  FS_REQUIRE(g && dsrc && y && mean && invstd && slab && C % 4 == 0 && C <= 1024 && Cg % 4 == 0 && coff % 4 == 0 && coff + C <= Cg);
  FS_REQUIRE(B > 0 && th > 0 && tw > 0 && Ho % th == 0 && Wo % tw == 0 && (Ho / th) % 2 == 0 && (Wo / tw) % 2 == 0);
  const long M = (long)B * th * tw;
  const int rpb = rows_per_block_for(M, C);
  hipLaunchKernelGGL(upsample_slice_bwd_bnsum_kernel, dim3(fs_bn_bwd_slabs(M, C)), dim3(256), slab_lds_bytes(C), stream, g, Ho, Wo, Cg, coff, dsrc,
                     th, tw, C, M, rpb, y, mean, invstd, slab);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_maxpool_fwd(const float* x, float* out, int* arg, int B, int H, int W, int C, int Ho, int Wo, int k, int stride, int pad,
                   hipStream_t stream) {
  FS_REQUIRE(x && out && arg && B > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && pad < k);
  FS_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1 && (long)H * W < 2147483647L);
  int blocks = cdiv((long)B * Ho * Wo * C, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, out, arg, B, H, W, C, Ho, Wo, k, stride, pad);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_maxpool_bwd(const float* dout, const int* arg, float* dx, int B, int H, int W, int C, int Ho, int Wo, int k, int stride,
                   int pad, hipStream_t stream) {
  FS_REQUIRE(dout && arg && dx && B > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && pad < k);
  int blocks = cdiv((long)B * H * W * C, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(maxpool_bwd_kernel, dim3(blocks), dim3(256), 0, stream, dout, arg, dx, B, H, W, C, Ho, Wo, k, stride, pad);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
// forward and backward are the same map (the mask depends only on the element index)
int fs_dropout(const float* x, float* out, long n, float drop_p, uint32_t drop_key, hipStream_t stream) {
  FS_REQUIRE(x && out && n > 0 && n < 4294967296L && drop_p > 0.f && drop_p < 1.f);
  const float scale = 1.0f / (float)(1.0 - (double)drop_p);
  const uint32_t thresh = (uint32_t)((double)drop_p * 4294967296.0);
  int blocks = cdiv(n, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(dropout_kernel, dim3(blocks), dim3(256), 0, stream, x, out, n, scale, thresh, drop_key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// rows per block of the two colsum kernels; the number of partial rows they write
static void colsum_plan(long M, int C, bool vec, int* rpb, int* nblk) {
  if (vec) *rpb = rows_per_block_for(M, C);
  else { int chunks = (int)((M + 255) / 256); if (chunks > 512) chunks = 512; *rpb = (int)((M + chunks - 1) / chunks); }
  *nblk = cdiv(M, *rpb);
}
// include/fovealseg.h: floats of scratch fs_colsum needs (per-row-block partial sums)
long fs_colsum_scratch_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rpb, nblk;
  colsum_plan(M, C, C % 4 == 0 && C <= 1024, &rpb, &nblk);
  return (long)nblk * C;
}
// accumulate != 0: the column sums are ADDED to out (a gradient-arena target), else out is overwritten.  Two launches: per-row-block
// partial sums into `scratch`, then their sum in block order (no atomics: the result does not depend on workgroup arrival order)
int fs_colsum(const float* x, long M, int C, float* out, int accumulate, float* scratch, hipStream_t stream) {
  FS_REQUIRE(x && out && scratch && M > 0 && C > 0);
  const bool vec = C % 4 == 0 && C <= 1024 && ((size_t)x & 15) == 0;
  int rpb, nblk;
  colsum_plan(M, C, C % 4 == 0 && C <= 1024, &rpb, &nblk);
  if (vec) {
    const int cw = C / 4;
    int rpi = 256 / cw; if (rpi < 1) rpi = 1;
    hipLaunchKernelGGL(colsum4_kernel, dim3(nblk), dim3(256), (size_t)rpi * C * sizeof(float), stream, x, M, C, rpb, scratch);
  } else {
    if (C % 4 == 0 && C <= 1024) colsum_plan(M, C, false, &rpb, &nblk);      // misaligned base: the scalar kernel (fewer blocks than planned)
    hipLaunchKernelGGL(colsum_kernel, dim3(cdiv(C, 256), nblk), dim3(256), 0, stream, x, M, C, rpb, scratch);
  }
  FS_LAUNCH_CHECK();
  return fs_slab_reduce(scratch, nblk, C, out, accumulate, stream);
}


int fs_avgpool_fwd(const float* x, int B, int HW, int C, float* out, hipStream_t stream) {
  FS_REQUIRE(x && out && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_fwd_kernel, dim3(cdiv((long)B * C, 256)), dim3(256), 0, stream, x, B, HW, C, out);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_avgpool_bwd(const float* dout, int B, int HW, int C, float* dx, hipStream_t stream) {
  FS_REQUIRE(dout && dx && B > 0 && HW > 0 && C > 0);
  hipLaunchKernelGGL(avgpool_bwd_kernel, dim3(cdiv((long)B * HW * C, 256)), dim3(256), 0, stream, dout, B, HW, C, dx);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // extern "C"
