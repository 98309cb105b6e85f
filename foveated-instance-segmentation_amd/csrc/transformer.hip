// SegFormer (Mix-Transformer) encoder pieces that are not convolutions / linears:
//   LayerNorm over channels, exact GELU, depthwise 3x3 conv (+bias), sequence-reduced multi-head attention
//   (<= 128 key/value tokens, head_dim 64), residual + DropPath.
// Replaces the ATen ops behind transformers==4.46.2 modeling_segformer (third-party; call sites
// models/segformer.py:2,9-11,33-37,88-100).  Tokens are rows of an NHWC tensor: (B, N=H*W, C).
#include "common.h"

bool fs_deterministic();      // conv.hip (include/fovealseg.h fs_set_deterministic)

namespace {

// ------------------------------------------------------------------------------------------
// LayerNorm: one wave per row (C <= 2048, C % 4 == 0)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float* __restrict__ y,
                                                            float* __restrict__ mean, float* __restrict__ rstd, long M, int C,
                                                            float eps) {
  const int lane = threadIdx.x & 63;
  const long row = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= M) return;
  const float* xr = x + row * C;
  f32x4 v[8];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 4 * lane + 256 * j;
    v[j] = f32x4{0, 0, 0, 0};
    if (c < C) { v[j] = *reinterpret_cast<const f32x4*>(xr + c); s += v[j].x + v[j].y + v[j].z + v[j].w; }
  }
  const float mu = wave_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < C) { const f32x4 d = v[j] - mu; q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w; }
  }
  const float rs = rsqrtf(wave_sum(q) / (float)C + eps);
  if (lane == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
      *reinterpret_cast<f32x4*>(y + row * C + c) = (v[j] - mu) * rs * g + b;
    }
  }
}

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));  per-block records of sum g*xhat (part_g) and sum g (part_b)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ part_g, float* __restrict__ part_b, long M, int C,
                                                            int rows_per_block, const float* __restrict__ addend) {
  extern __shared__ float sm[];          // [4 waves][2][C] partials of dgamma / dbeta
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  f32x4 ag[8], ab[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { ag[j] = f32x4{0, 0, 0, 0}; ab[j] = f32x4{0, 0, 0, 0}; }
  for (long row = r0 + wv; row < r1; row += 4) {
    const float mu = mean[row], rs = rstd[row];
    f32x4 gg[8], xh[8];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 4 * lane + 256 * j;
      gg[j] = f32x4{0, 0, 0, 0}; xh[j] = f32x4{0, 0, 0, 0};
      if (c < C) {
        const f32x4 go = *reinterpret_cast<const f32x4*>(g + row * C + c);
        xh[j] = (*reinterpret_cast<const f32x4*>(x + row * C + c) - mu) * rs;
        ag[j] += go * xh[j]; ab[j] += go;
        gg[j] = go * *reinterpret_cast<const f32x4*>(gamma + c);
        s1 += gg[j].x + gg[j].y + gg[j].z + gg[j].w;
        const f32x4 t = gg[j] * xh[j];
        s2 += t.x + t.y + t.z + t.w;
      }
    }
    const float m1 = wave_sum(s1) / (float)C, m2 = wave_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = 4 * lane + 256 * j;
      if (c < C) {
        f32x4 d = rs * (gg[j] - m1 - xh[j] * m2);
        if (addend != nullptr) d += *reinterpret_cast<const f32x4*>(addend + row * C + c);      // the other gradient of x (the residual branch's)
        *reinterpret_cast<f32x4*>(dx + row * C + c) = d;
      }
    }
  }
  // sm = [4 waves][2 C]: every wave stores its partial sums, the block adds them in wave order and writes its record to
  // part_g[block][C] / part_b[block][C]; fs_slab_reduce adds the records in block order (no atomics: nothing depends on arrival order)
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < C) {
      *reinterpret_cast<f32x4*>(&sm[wv * 2 * C + c]) = ag[j];
      *reinterpret_cast<f32x4*>(&sm[wv * 2 * C + C + c]) = ab[j];
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    part_g[(long)blockIdx.x * C + c] = ((sm[c] + sm[2 * C + c]) + sm[4 * C + c]) + sm[6 * C + c];
    part_b[(long)blockIdx.x * C + c] = ((sm[C + c] + sm[3 * C + c]) + sm[5 * C + c]) + sm[7 * C + c];
  }
}

// Sixteen lanes per row (C <= 320): the Mix-Transformer widths are 64 .. 512, where a wave per row leaves 3/4 (C = 64) to 3/8 (C = 320,
// second pass) of its lanes idle -- 0.6 TB/s on the 160 x 160 stage.  Lane l of a group owns the float4s at 4 l + 64 j, j < NJ = ceil(C / 64):
// a group reads 256 contiguous bytes per j, a wave works on four rows, and the row reductions are four DPP-width shuffles.
__device__ __forceinline__ float group16_sum(float v) {
#pragma unroll
  for (int o = 8; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}

template <int NJ>
__global__ __launch_bounds__(256) void layernorm_fwd16_kernel(const float* __restrict__ x, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, float* __restrict__ y,
                                                              float* __restrict__ mean, float* __restrict__ rstd, long M, int C, float eps) {
  const int l16 = threadIdx.x & 15;
  const long row = ((long)blockIdx.x * 256 + threadIdx.x) >> 4;
  const bool rok = row < M;                     // (lanes of a missing row stay for the shuffles)
  const float* xr = x + row * C;
  f32x4 v[NJ];
  float s = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4 * l16 + 64 * j;
    v[j] = f32x4{0, 0, 0, 0};
    if (rok && c < C) { v[j] = *reinterpret_cast<const f32x4*>(xr + c); s += v[j].x + v[j].y + v[j].z + v[j].w; }
  }
  const float mu = group16_sum(s) / (float)C;
  float q = 0.f;
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4 * l16 + 64 * j;
    if (c < C) { const f32x4 d = v[j] - mu; q += d.x * d.x + d.y * d.y + d.z * d.z + d.w * d.w; }
  }
  const float rs = rsqrtf(group16_sum(q) / (float)C + eps);
  if (!rok) return;
  if (l16 == 0) { mean[row] = mu; rstd[row] = rs; }
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4 * l16 + 64 * j;
    if (c < C) {
      const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c), b = *reinterpret_cast<const f32x4*>(beta + c);
      *reinterpret_cast<f32x4*>(y + row * C + c) = (v[j] - mu) * rs * g + b;
    }
  }
}

template <int NJ>
__global__ __launch_bounds__(256) void layernorm_bwd16_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                              const float* __restrict__ gamma, const float* __restrict__ mean,
                                                              const float* __restrict__ rstd, float* __restrict__ dx,
                                                              float* __restrict__ part_g, float* __restrict__ part_b, long M, int C,
                                                              int rows_per_block, const float* __restrict__ addend) {
  extern __shared__ float sm[];          // [4 waves][2][C] partials of dgamma / dbeta
  const int l16 = threadIdx.x & 15, grp = threadIdx.x >> 4;
  const long r0 = (long)blockIdx.x * rows_per_block;
  long r1 = r0 + rows_per_block; if (r1 > M) r1 = M;
  f32x4 ag[NJ], ab[NJ], gm[NJ];
#pragma unroll
  for (int j = 0; j < NJ; ++j) {
    const int c = 4 * l16 + 64 * j;
    ag[j] = f32x4{0, 0, 0, 0}; ab[j] = f32x4{0, 0, 0, 0};
    gm[j] = c < C ? *reinterpret_cast<const f32x4*>(gamma + c) : f32x4{0, 0, 0, 0};
  }
  for (long rb = r0; rb < r1; rb += 16) {       // (uniform trip count per wave: the shuffles need all four groups)
    const long row = rb + grp;
    const bool rok = row < r1;
    const float mu = rok ? mean[row] : 0.f, rs = rok ? rstd[row] : 0.f;
    f32x4 gg[NJ], xh[NJ];
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4 * l16 + 64 * j;
      gg[j] = f32x4{0, 0, 0, 0}; xh[j] = f32x4{0, 0, 0, 0};
      if (rok && c < C) {
        const f32x4 go = *reinterpret_cast<const f32x4*>(g + row * C + c);
        xh[j] = (*reinterpret_cast<const f32x4*>(x + row * C + c) - mu) * rs;
        ag[j] += go * xh[j]; ab[j] += go;
        gg[j] = go * gm[j];
        s1 += gg[j].x + gg[j].y + gg[j].z + gg[j].w;
        const f32x4 t = gg[j] * xh[j];
        s2 += t.x + t.y + t.z + t.w;
      }
    }
    const float m1 = group16_sum(s1) / (float)C, m2 = group16_sum(s2) / (float)C;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4 * l16 + 64 * j;
      if (rok && c < C) {
        f32x4 d = rs * (gg[j] - m1 - xh[j] * m2);
        if (addend != nullptr) d += *reinterpret_cast<const f32x4*>(addend + row * C + c);
        *reinterpret_cast<f32x4*>(dx + row * C + c) = d;
      }
    }
  }
  // the four groups of a wave hold partials of the same channels: add them across the wave first (lanes l, l ^ 16, l ^ 32, l ^ 48)
#pragma unroll
  for (int j = 0; j < NJ; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      ag[j][e] += __shfl_xor(ag[j][e], 16, 64); ag[j][e] += __shfl_xor(ag[j][e], 32, 64);
      ab[j][e] += __shfl_xor(ab[j][e], 16, 64); ab[j][e] += __shfl_xor(ab[j][e], 32, 64);
    }
  if ((threadIdx.x & 63) < 16) {
    const int wv = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int c = 4 * l16 + 64 * j;
      if (c < C) {
        *reinterpret_cast<f32x4*>(&sm[wv * 2 * C + c]) = ag[j];
        *reinterpret_cast<f32x4*>(&sm[wv * 2 * C + C + c]) = ab[j];
      }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    part_g[(long)blockIdx.x * C + c] = ((sm[c] + sm[2 * C + c]) + sm[4 * C + c]) + sm[6 * C + c];
    part_b[(long)blockIdx.x * C + c] = ((sm[C + c] + sm[3 * C + c]) + sm[5 * C + c]) + sm[7 * C + c];
  }
}

// ------------------------------------------------------------------------------------------
// exact GELU
// ------------------------------------------------------------------------------------------
// thresh != 0: the nn.Dropout that follows the activation (MixFFN) in the same pass -- element hash of fs_dropout on the output's index
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4, float drop_scale,
                                                       uint32_t thresh, uint32_t key) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    const uint32_t keep = thresh != 0u ? fs_dropout_keep4((uint32_t)(4 * i), key, thresh) : 15u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752440f));
      if (thresh != 0u) v[j] = ((keep >> j) & 1u) ? v[j] * drop_scale : 0.f;
    }
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                       float* __restrict__ dx, long n4, float drop_scale, uint32_t thresh, uint32_t key) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 d = reinterpret_cast<const f32x4*>(g)[i];
    const uint32_t keep = thresh != 0u ? fs_dropout_keep4((uint32_t)(4 * i), key, thresh) : 15u;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cdf = 0.5f * (1.f + erff(v[j] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * v[j] * v[j]);
      if (thresh != 0u) d[j] = ((keep >> j) & 1u) ? d[j] * drop_scale : 0.f;
      d[j] *= cdf + v[j] * pdf;
    }
    reinterpret_cast<f32x4*>(dx)[i] = d;
  }
}

// ------------------------------------------------------------------------------------------
// depthwise 3x3 conv, stride 1, pad 1, NHWC; weights logical (C,1,3,3) = [c][9]
// flip = 0: y = sum_t w[c][t] x[p + t] (+ bias);  flip = 1: input gradient (taps mirrored, no bias)
// HBM-bound (one read of x, one write of y).  Round 2's kernel issued 9 neighbour loads per output through L1 and ran at a fifth
// of the roof; here a thread owns 4 channels x DW_RY output rows and WALKS along the row with a 3-column register window: 6 new
// loads per 4 outputs.  Consecutive lanes = consecutive channel quads (1 KB per pixel row of a wave).
// ------------------------------------------------------------------------------------------
constexpr int DW_RY = 4;       // output rows per thread
constexpr int DW_SEG = 16;     // output columns per thread

__global__ __launch_bounds__(256) void dwconv3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                      int C, int flip) {
  const int cw = C >> 2;
  const int rg = (H + DW_RY - 1) / DW_RY, sg = (W + DW_SEG - 1) / DW_SEG;
  const long items = (long)B * rg * sg;
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = (int)(gt % cw);
  const long item = gt / cw;
  if (item >= items) return;
  const int c = 4 * cq;
  const int s_ = (int)(item % sg), r_ = (int)((item / sg) % rg), b = (int)(item / ((long)sg * rg));
  const int y0 = r_ * DW_RY, x0 = s_ * DW_SEG;
  const int x1 = x0 + DW_SEG < W ? x0 + DW_SEG : W;
  f32x4 wv[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const int tt = flip ? 8 - t : t;
#pragma unroll
    for (int e = 0; e < 4; ++e) wv[t][e] = w[(c + e) * 9 + tt];
  }
  const f32x4 bv = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0, 0, 0, 0};
  const float* xb = x + (long)b * H * W * C + c;
  auto col = [&](int ix, f32x4 (&v)[DW_RY + 2]) {
#pragma unroll
    for (int r = 0; r < DW_RY + 2; ++r) {
      const int iy = y0 - 1 + r;
      const bool ok = ix >= 0 && ix < W && iy >= 0 && iy < H;
      v[r] = ok ? *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + ix) * C) : f32x4{0, 0, 0, 0};
    }
  };
  f32x4 v0[DW_RY + 2], v1[DW_RY + 2], v2[DW_RY + 2];
  col(x0 - 1, v0);
  col(x0, v1);
  for (int ix = x0; ix < x1; ++ix) {
    col(ix + 1, v2);
#pragma unroll
    for (int r = 0; r < DW_RY; ++r) {
      if (y0 + r >= H) break;
      f32x4 acc = bv;
#pragma unroll
      for (int ty = 0; ty < 3; ++ty) acc += v0[r + ty] * wv[3 * ty] + v1[r + ty] * wv[3 * ty + 1] + v2[r + ty] * wv[3 * ty + 2];
      *reinterpret_cast<f32x4*>(y + (((long)b * H + y0 + r) * W + ix) * C + c) = acc;
    }
#pragma unroll
    for (int r = 0; r < DW_RY + 2; ++r) { v0[r] = v1[r]; v1[r] = v2[r]; }
  }
}

// dw[c][t] = sum_pix x[pix + t][c] * dy[pix][c].  Same walk (10 loads per 4 pixels instead of 40); a thread loops over the items
// item0, item0 + nitem_lanes, ... and writes its 9 x 4 partial sums to slab[item lane][9][C] (coalesced, no atomics, no memset);
// dwconv3_wgrad_reduce_kernel adds the slab rows into dw.
// nt = 10: the bias gradient (column sums of dy, which this kernel reads exactly once anyway) rides along as a tenth slab plane.
__global__ __launch_bounds__(256) void dwconv3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                           float* __restrict__ slab, int B, int H, int W, int C, int nlanes, int nt) {
  const int cw = C >> 2;
  const int rg = (H + DW_RY - 1) / DW_RY, sg = (W + DW_SEG - 1) / DW_SEG;
  const long items = (long)B * rg * sg;
  const long gt = (long)blockIdx.x * blockDim.x + threadIdx.x;
  const int cq = (int)(gt % cw);
  const long lane0 = gt / cw;
  if (lane0 >= nlanes) return;
  const int c = 4 * cq;
  f32x4 acc[9], accb = f32x4{0, 0, 0, 0};
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (long item = lane0; item < items; item += nlanes) {
    const int s_ = (int)(item % sg), r_ = (int)((item / sg) % rg), b = (int)(item / ((long)sg * rg));
    const int y0 = r_ * DW_RY, x0 = s_ * DW_SEG;
    const int x1 = x0 + DW_SEG < W ? x0 + DW_SEG : W;
    const float* xb = x + (long)b * H * W * C + c;
    const float* gb = dy + (long)b * H * W * C + c;
    auto col = [&](int ix, f32x4 (&v)[DW_RY + 2]) {
#pragma unroll
      for (int r = 0; r < DW_RY + 2; ++r) {
        const int iy = y0 - 1 + r;
        const bool ok = ix >= 0 && ix < W && iy >= 0 && iy < H;
        v[r] = ok ? *reinterpret_cast<const f32x4*>(xb + ((long)iy * W + ix) * C) : f32x4{0, 0, 0, 0};
      }
    };
    f32x4 v0[DW_RY + 2], v1[DW_RY + 2], v2[DW_RY + 2];
    col(x0 - 1, v0);
    col(x0, v1);
    for (int ix = x0; ix < x1; ++ix) {
      col(ix + 1, v2);
#pragma unroll
      for (int r = 0; r < DW_RY; ++r) {
        const bool ok = y0 + r < H;
        const f32x4 g = ok ? *reinterpret_cast<const f32x4*>(gb + ((long)(y0 + r) * W + ix) * C) : f32x4{0, 0, 0, 0};
        accb += g;
#pragma unroll
        for (int ty = 0; ty < 3; ++ty) {
          acc[3 * ty] += g * v0[r + ty];
          acc[3 * ty + 1] += g * v1[r + ty];
          acc[3 * ty + 2] += g * v2[r + ty];
        }
      }
#pragma unroll
      for (int r = 0; r < DW_RY + 2; ++r) { v0[r] = v1[r]; v1[r] = v2[r]; }
    }
  }
  float* dst = slab + (long)lane0 * nt * C + c;
#pragma unroll
  for (int t = 0; t < 9; ++t) *reinterpret_cast<f32x4*>(dst + (long)t * C) = acc[t];
  if (nt == 10) *reinterpret_cast<f32x4*>(dst + 9L * C) = accb;
}

// dw[c][t] (+)= sum over slab rows of slab[row][t][c]: block = 32 consecutive (t, c) elements x 8 row groups
__global__ __launch_bounds__(256) void dwconv3_wgrad_reduce_kernel(const float* __restrict__ slab, int nlanes, int C, float* __restrict__ dw,
                                                                  int accumulate, int nt, float* __restrict__ db, int accumulate_b) {
  __shared__ float red[8][32];
  const int e = threadIdx.x & 31, rgp = threadIdx.x >> 5;
  const int i = blockIdx.x * 32 + e;                          // i = t * C + c: 128-byte segments per slab row
  float s = 0.f;
  if (i < nt * C)
    for (int r = rgp; r < nlanes; r += 8) s += slab[(long)r * nt * C + i];
  red[rgp][e] = s;
  __syncthreads();
  if (rgp == 0 && i < nt * C) {
#pragma unroll
    for (int k = 1; k < 8; ++k) s += red[k][e];
    const int t = i / C, c = i - t * C;
    float* d = t < 9 ? dw + c * 9 + t : db + c;
    *d = ((t < 9 ? accumulate : accumulate_b) ? *d : 0.f) + s;
  }
}

// ------------------------------------------------------------------------------------------
// residual + DropPath: out = x + keep_b * y / (1-p)   (p = 0: plain add); keep per sample from the hash
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void residual_droppath_kernel(const float* __restrict__ x, const float* __restrict__ y,
                                                                float* __restrict__ out, long n4, long per_sample4, float scale,
                                                                uint32_t thresh, uint32_t key) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint32_t b = (uint32_t)(i / per_sample4);
    const float k = (thresh == 0u || fs_dropout_keep(b, key, thresh)) ? scale : 0.f;
    f32x4 v = reinterpret_cast<const f32x4*>(y)[i] * k;
    if (x != nullptr) v += reinterpret_cast<const f32x4*>(x)[i];
    reinterpret_cast<f32x4*>(out)[i] = v;
  }
}

// Backward of y = x + DropPath(Dropout(z)) with respect to z in one pass: dz = Dropout-mask(DropPath-scale_b * g), the two steps in the
// order the separate passes apply them (fs_residual_droppath with x = NULL, then fs_dropout on the result)
__global__ __launch_bounds__(256) void droppath_dropout_bwd_kernel(const float* __restrict__ g, float* __restrict__ dz, long n4, long per_sample4,
                                                                   float dp_scale, uint32_t dp_thresh, uint32_t dp_key, float drop_scale,
                                                                   uint32_t drop_thresh, uint32_t drop_key) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const uint32_t b = (uint32_t)(i / per_sample4);
    const float k = (dp_thresh == 0u || fs_dropout_keep(b, dp_key, dp_thresh)) ? dp_scale : 0.f;
    f32x4 v = reinterpret_cast<const f32x4*>(g)[i] * k;
    if (drop_thresh != 0u) {
      const uint32_t keep = fs_dropout_keep4((uint32_t)(4 * i), drop_key, drop_thresh);
#pragma unroll
      for (int j = 0; j < 4; ++j) v[j] = ((keep >> j) & 1u) ? v[j] * drop_scale : 0.f;
    }
    reinterpret_cast<f32x4*>(dz)[i] = v;
  }
}

// ------------------------------------------------------------------------------------------
// unfold / fold: a convolution whose filter has more taps than the aligned conv kernels take (7x7 patch embedding on 3 channels, the 8x8
// stride-8 sequence-reduction conv) as rows of patches [B*Ho*Wo][Kp], Kp = k*k*C padded to a multiple of 4 (>= 16), element order (r, s, c)
// = the RSCK weight's row order -- the layer is then ONE linear layer on the split-precision GEMM kernels (forward, input gradient, weight
// gradient) instead of the generic fp32 kernels (configs[3]: 13 ms of a 228 ms step).  fold is the adjoint (input gradient of unfold).
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void unfold_kernel(const float* __restrict__ x, float* __restrict__ col, int B, int H, int W, int C, int k,
                                                     int stride, int pad, int Ho, int Wo, int Kp, long total4) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
    const int kq = Kp >> 2;
    const long row = t / kq;
    const int e0 = (int)(t - row * kq) * 4;
    const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((long)Wo * Ho));
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int e = e0 + j;
      if (e < k * k * C) {
        const int c = e % C, rs = e / C, s_ = rs % k, r_ = rs / k;
        const int iy = oy * stride - pad + r_, ix = ox * stride - pad + s_;
        if (iy >= 0 && iy < H && ix >= 0 && ix < W) v[j] = x[(((long)b * H + iy) * W + ix) * C + c];
      }
    }
    reinterpret_cast<f32x4*>(col)[t] = v;
  }
}
// dx[b][y][x][c] = sum over the taps (r, s) whose output pixel exists of col[(b, oy, ox)][(r, s, c)]
__global__ __launch_bounds__(256) void fold_kernel(const float* __restrict__ col, float* __restrict__ dx, int B, int H, int W, int C, int k,
                                                   int stride, int pad, int Ho, int Wo, int Kp, long total) {
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total; t += (long)gridDim.x * 256) {
    const int c = (int)(t % C);
    const long pix = t / C;
    const int ix = (int)(pix % W), iy = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    float acc = 0.f;
    for (int r_ = 0; r_ < k; ++r_) {
      const int ny = iy + pad - r_;
      if (ny < 0 || ny % stride) continue;
      const int oy = ny / stride;
      if (oy >= Ho) continue;
      for (int s_ = 0; s_ < k; ++s_) {
        const int nx = ix + pad - s_;
        if (nx < 0 || nx % stride) continue;
        const int ox = nx / stride;
        if (ox >= Wo) continue;
        acc += col[(((long)b * Ho + oy) * Wo + ox) * Kp + (r_ * k + s_) * C + c];
      }
    }
    dx[t] = acc;
  }
}

// Non-overlapping patches (stride = k, no padding, C a multiple of 4: the sequence-reduction convs of the Mix-Transformer, 46 of the 50
// unfold / fold pairs of a configs[3] step): every pixel belongs to exactly one patch element, so both directions are pure permutations of
// float4s -- one load, one store, a handful of integer divisions per FOUR elements (the general kernels above walk the k x k taps per
// scalar element with a division and a remainder each: 73 us for a 33 MB fold).
__global__ __launch_bounds__(256) void unfold_patch_kernel(const float* __restrict__ x, float* __restrict__ col, int H, int W, int C, int k,
                                                           int Ho, int Wo, long total4) {
  const int kq = (k * k * C) >> 2, cq = C >> 2;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
    const long row = t / kq;
    const int q = (int)(t - row * kq);
    const int rs = q / cq, c4 = q - rs * cq;
    const int r_ = rs / k, s_ = rs - r_ * k;
    const int ox = (int)(row % Wo);
    const long t2 = row / Wo;
    const int oy = (int)(t2 % Ho), b = (int)(t2 / Ho);
    reinterpret_cast<f32x4*>(col)[t] = *reinterpret_cast<const f32x4*>(x + (((long)b * H + oy * k + r_) * W + ox * k + s_) * C + 4 * c4);
  }
}
__global__ __launch_bounds__(256) void fold_patch_kernel(const float* __restrict__ col, float* __restrict__ dx, int H, int W, int C, int k,
                                                         int Ho, int Wo, long total4) {
  const int cq = C >> 2;
  const long Kp = (long)k * k * C;
  for (long t = (long)blockIdx.x * 256 + threadIdx.x; t < total4; t += (long)gridDim.x * 256) {
    const long pix = t / cq;
    const int c4 = (int)(t - pix * cq);
    const int ix = (int)(pix % W);
    const long t2 = pix / W;
    const int iy = (int)(t2 % H), b = (int)(t2 / H);
    const int oy = iy / k, r_ = iy - oy * k, ox = ix / k, s_ = ix - ox * k;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};                       // (pixels past the last whole patch belong to no output)
    if (oy < Ho && ox < Wo) v = *reinterpret_cast<const f32x4*>(col + (((long)b * Ho + oy) * Wo + ox) * Kp + (long)(r_ * k + s_) * C + 4 * c4);
    reinterpret_cast<f32x4*>(dx)[t] = v;
  }
}

// ------------------------------------------------------------------------------------------
// Attention on the matrix cores, exact fp32: S = (q*scale) K^T, P = softmax(S), P~ = dropout(P), O = P~ V per (batch, head),
// head_dim 64, any key count (the sequence-reduced keys of SegFormer: 100 at 80x80, 400 at 160x160), keys streamed through
// LDS in chunks of 64 with an online softmax.  Products run on v_mfma_f32_32x32x2_f32 (fp32 in, fp32 accumulate: bit-for-bit
// an fmaf chain), so the result sits at the same error level as the fp32 oracle.
//
// Operand maps of v_mfma_f32_32x32x2_f32 (cdna_hip_programming.md): lane l supplies A[i = l&31][k = l>>5] and
// B[k = l>>5][j = l&31]; result element (row = (reg&3) + 8*(reg>>2) + 4*(l>>5), col = l&31) in register reg.
//   * "row-major operand" (Q, K, dO, V as the contraction-over-head-dim operand, P as the contraction-over-keys operand):
//     lane (r, h) reads the float2 at [r][4j + 2h] -> the two MFMAs of step j contract k = 4j + 2h and 4j + 2h + 1.  Both
//     operands of a product use the same map, so the permuted k order is harmless.  LDS rows are 66 floats: the 32 lanes of
//     a ds_read_b64 group hit 32 distinct even banks.
//   * "k-major operand" (V in P~V, K in dS K, dO / Q in the dK / dV products): lane (r, h) reads [k][32*dt + r], consecutive
//     across the group.
//   * dV = P~^T dO and dK = dS^T Q contract over the ROW index of the P~ / dS tiles, which is the register index of the
//     accumulator layout: register i of lane (r, h) IS A[key r][query (i&3) + 8*(i>>2) + 4h], no transpose, no LDS.
// q (B,N,heads*64), k/v (B,Nk,heads*64), o (B,N,heads*64); lse = B*heads*N floats.
// ------------------------------------------------------------------------------------------
constexpr int HD = 64, ALD = 66, KC = 64, QW = 32;

__device__ __forceinline__ float half_max(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float half_sum(float v) {
#pragma unroll
  for (int o = 16; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
// LDS written by one lane of a wave and read by another lane of the SAME wave: order the accesses, no workgroup barrier
__device__ __forceinline__ void wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}
// nrows x 64 floats (global rows `stride` floats apart, rows >= nvalid read as zero) -> LDS rows of ALD floats, times mul
__device__ __forceinline__ void stage_rows(const float* __restrict__ src, long stride, int nvalid, float* lds, int nrows, int t, int nt, float mul) {
  for (int i = t; i < nrows * 16; i += nt) {
    const int rr = i >> 4, c4 = (i & 15) * 4;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (rr < nvalid) v = *reinterpret_cast<const f32x4*>(src + (long)rr * stride + c4) * mul;
    float2* d = reinterpret_cast<float2*>(&lds[rr * ALD + c4]);
    d[0] = float2{v.x, v.y};
    d[1] = float2{v.z, v.w};
  }
}
__device__ __forceinline__ f32x16 mfma32(float a, float b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, c, 0, 0, 0); }
__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }

__global__ __launch_bounds__(256, 2) void attn_mfma_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                            const float* __restrict__ v, float* __restrict__ o, float* __restrict__ lse,
                                                            int N, int Nk, int heads, float scale, float drop_scale, uint32_t thresh,
                                                            uint32_t key) {
  __shared__ __attribute__((aligned(16))) float Ks[KC * ALD], Vs[KC * ALD], Ps[4][QW * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const int q0 = blockIdx.x * 128 + wave * QW;
  const float* qb = q + (long)b * N * C + hd * HD;
  const float* kb = k + (long)b * Nk * C + hd * HD;
  const float* vb = v + (long)b * Nk * C + hd * HD;
  float* Pw = Ps[wave];
  stage_rows(qb + (long)q0 * C, C, N - q0, Pw, QW, lane, 64, scale);
  wave_sync();
  float2 qa[16];
#pragma unroll
  for (int j = 0; j < 16; ++j) qa[j] = *reinterpret_cast<const float2*>(&Pw[r * ALD + 4 * j + 2 * h]);
  f32x16 O[2];
  float m[16], l[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) { O[0][i] = 0.f; O[1][i] = 0.f; m[i] = -INFINITY; l[i] = 0.f; }
  const int nchunk = (Nk + KC - 1) / KC;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                           // every wave is done with the previous chunk (and with its Q staging)
    stage_rows(kb + (long)c * KC * C, C, Nk - c * KC, Ks, KC, tid, 256, 1.f);
    stage_rows(vb + (long)c * KC * C, C, Nk - c * KC, Vs, KC, tid, 256, 1.f);
    __syncthreads();
    f32x16 S[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 kf = *reinterpret_cast<const float2*>(&Ks[(32 * t + r) * ALD + 4 * j + 2 * h]);
        S[t] = mfma32(qa[j].x, kf.x, S[t]);
        S[t] = mfma32(qa[j].y, kf.y, S[t]);
      }
    }
    const int key0 = c * KC + r, key1 = key0 + 32;
    const bool ok0 = key0 < Nk, ok1 = key1 < Nk;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const float s0 = ok0 ? S[0][i] : -INFINITY, s1 = ok1 ? S[1][i] : -INFINITY;
      const float mn = fmaxf(m[i], half_max(fmaxf(s0, s1)));
      const float alpha = expf(m[i] - mn);
      float p0 = expf(s0 - mn), p1 = expf(s1 - mn);
      l[i] = l[i] * alpha + half_sum(p0 + p1);
      m[i] = mn;
      O[0][i] *= alpha; O[1][i] *= alpha;
      const int rl = acc_row(i, h);
      if (thresh != 0u) {
        const uint32_t e = (uint32_t)(((long)bh * N + q0 + rl) * Nk);
        p0 = fs_dropout_keep(e + (uint32_t)key0, key, thresh) ? p0 * drop_scale : 0.f;
        p1 = fs_dropout_keep(e + (uint32_t)key1, key, thresh) ? p1 * drop_scale : 0.f;
      }
      Pw[rl * ALD + r] = p0;
      Pw[rl * ALD + 32 + r] = p1;
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 pa = *reinterpret_cast<const float2*>(&Pw[r * ALD + 4 * j + 2 * h]);
      const int kk = 4 * j + 2 * h;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        O[dt] = mfma32(pa.x, Vs[kk * ALD + 32 * dt + r], O[dt]);
        O[dt] = mfma32(pa.y, Vs[(kk + 1) * ALD + 32 * dt + r], O[dt]);
      }
    }
    wave_sync();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = q0 + acc_row(i, h);
    if (row >= N) continue;
    const float inv = 1.f / l[i];
    float* orow = o + ((long)b * N + row) * C + hd * HD;
    orow[r] = O[0][i] * inv;
    orow[32 + r] = O[1][i] * inv;
    if (r == 0) lse[(long)bh * N + row] = m[i] + logf(l[i]);
  }
}

// D[bh][row] = sum_d dO[row][d] * O[row][d]  (= sum_j P~_j dP~_j): 16 lanes per row
__global__ __launch_bounds__(256) void attn_rowdot_kernel(const float* __restrict__ go, const float* __restrict__ o, float* __restrict__ D,
                                                          int B, int N, int heads) {
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const long rowid = t >> 4;                       // (b*heads + hd)*N + row
  const int c4 = (int)(t & 15) * 4;
  const long total = (long)B * heads * N;
  float s = 0.f;
  if (rowid < total) {
    const int bh = (int)(rowid / N), row = (int)(rowid - (long)bh * N);
    const int b = bh / heads, hd = bh - b * heads;
    const long off = ((long)b * N + row) * heads * HD + hd * HD + c4;
    const f32x4 a = *reinterpret_cast<const f32x4*>(go + off), c = *reinterpret_cast<const f32x4*>(o + off);
    s = a.x * c.x + a.y * c.y + a.z * c.z + a.w * c.w;
  }
#pragma unroll
  for (int w = 8; w > 0; w >>= 1) s += __shfl_xor(s, w, 64);
  if (rowid < total && (t & 15) == 0) D[rowid] = s;
}

// dQ = dS K with dS = P * (mask * dP~ - D) * scale: a workgroup owns 128 query rows and streams the keys (no atomics)
__global__ __launch_bounds__(256) void attn_mfma_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                               const float* __restrict__ v, const float* __restrict__ go,
                                                               const float* __restrict__ lse, const float* __restrict__ D,
                                                               float* __restrict__ dq, int N, int Nk, int heads, float scale,
                                                               float drop_scale, uint32_t thresh, uint32_t key) {
  __shared__ __attribute__((aligned(16))) float Ks[KC * ALD], Vs[KC * ALD], Ps[4][QW * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const int q0 = blockIdx.x * 128 + wave * QW;
  const float* qb = q + (long)b * N * C + hd * HD;
  const float* gb = go + (long)b * N * C + hd * HD;
  const float* kb = k + (long)b * Nk * C + hd * HD;
  const float* vb = v + (long)b * Nk * C + hd * HD;
  float* Pw = Ps[wave];
  float2 qa[16], ga[16];
  stage_rows(qb + (long)q0 * C, C, N - q0, Pw, QW, lane, 64, scale);
  wave_sync();
#pragma unroll
  for (int j = 0; j < 16; ++j) qa[j] = *reinterpret_cast<const float2*>(&Pw[r * ALD + 4 * j + 2 * h]);
  wave_sync();
  stage_rows(gb + (long)q0 * C, C, N - q0, Pw, QW, lane, 64, 1.f);
  wave_sync();
#pragma unroll
  for (int j = 0; j < 16; ++j) ga[j] = *reinterpret_cast<const float2*>(&Pw[r * ALD + 4 * j + 2 * h]);
  float L[16], Dr[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = q0 + acc_row(i, h);
    L[i] = row < N ? lse[(long)bh * N + row] : INFINITY;        // p = exp(s - inf) = 0 for rows past the end
    Dr[i] = row < N ? D[(long)bh * N + row] : 0.f;
  }
  f32x16 dQ[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dQ[0][i] = 0.f; dQ[1][i] = 0.f; }
  const int nchunk = (Nk + KC - 1) / KC;
  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    stage_rows(kb + (long)c * KC * C, C, Nk - c * KC, Ks, KC, tid, 256, 1.f);
    stage_rows(vb + (long)c * KC * C, C, Nk - c * KC, Vs, KC, tid, 256, 1.f);
    __syncthreads();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 kf = *reinterpret_cast<const float2*>(&Ks[(32 * t + r) * ALD + 4 * j + 2 * h]);
        const float2 vf = *reinterpret_cast<const float2*>(&Vs[(32 * t + r) * ALD + 4 * j + 2 * h]);
        S = mfma32(qa[j].x, kf.x, S);
        dP = mfma32(ga[j].x, vf.x, dP);
        S = mfma32(qa[j].y, kf.y, S);
        dP = mfma32(ga[j].y, vf.y, dP);
      }
      const int keyi = c * KC + 32 * t + r;
      const bool kok = keyi < Nk;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = acc_row(i, h);
        const float p = kok ? expf(S[i] - L[i]) : 0.f;
        float mk = 1.f;
        if (thresh != 0u) mk = fs_dropout_keep((uint32_t)(((long)bh * N + q0 + rl) * Nk) + (uint32_t)keyi, key, thresh) ? drop_scale : 0.f;
        Pw[rl * ALD + 32 * t + r] = p * (mk * dP[i] - Dr[i]) * scale;
      }
    }
    wave_sync();
#pragma unroll
    for (int j = 0; j < 16; ++j) {
      const float2 da = *reinterpret_cast<const float2*>(&Pw[r * ALD + 4 * j + 2 * h]);
      const int kk = 4 * j + 2 * h;
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        dQ[dt] = mfma32(da.x, Ks[kk * ALD + 32 * dt + r], dQ[dt]);
        dQ[dt] = mfma32(da.y, Ks[(kk + 1) * ALD + 32 * dt + r], dQ[dt]);
      }
    }
    wave_sync();
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    const int row = q0 + acc_row(i, h);
    if (row >= N) continue;
    float* drow = dq + ((long)b * N + row) * C + hd * HD;
    drow[r] = dQ[0][i];
    drow[32 + r] = dQ[1][i];
  }
}

// dK = dS^T Q, dV = P~^T dO: a workgroup owns 64 keys and streams query blocks [qb_begin, qb_end) of 128 rows; the four
// waves' partial tiles are summed through LDS in a fixed order.  atomics != 0 (several query ranges per key chunk): the
// result is added to zero-initialised dk / dv.
__global__ __launch_bounds__(256) void attn_mfma_bwd_dkv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ go,
                                                                const float* __restrict__ lse, const float* __restrict__ D,
                                                                float* __restrict__ dk, float* __restrict__ dv, int N, int Nk, int heads,
                                                                float scale, float drop_scale, uint32_t thresh, uint32_t key,
                                                                int blocks_per_split, int atomics) {
  __shared__ __attribute__((aligned(16))) float Ks[KC * ALD], Vs[KC * ALD], Qs[4][QW * ALD], Gs[4][QW * ALD];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const float* qb = q + (long)b * N * C + hd * HD;
  const float* gb = go + (long)b * N * C + hd * HD;
  const float* kb = k + (long)b * Nk * C + hd * HD;
  const float* vb = v + (long)b * Nk * C + hd * HD;
  stage_rows(kb + (long)c * KC * C, C, Nk - c * KC, Ks, KC, tid, 256, 1.f);
  stage_rows(vb + (long)c * KC * C, C, Nk - c * KC, Vs, KC, tid, 256, 1.f);
  __syncthreads();
  float* Qw = Qs[wave];
  float* Gw = Gs[wave];
  f32x16 dK[2][2], dV[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int dt = 0; dt < 2; ++dt)
#pragma unroll
      for (int i = 0; i < 16; ++i) { dK[t][dt][i] = 0.f; dV[t][dt][i] = 0.f; }
  const int nqb = (N + 127) / 128;
  const int qb_begin = blockIdx.z * blocks_per_split;
  const int qb_end = qb_begin + blocks_per_split < nqb ? qb_begin + blocks_per_split : nqb;
  for (int qblk = qb_begin; qblk < qb_end; ++qblk) {
    const int q0 = qblk * 128 + wave * QW;
    stage_rows(qb + (long)q0 * C, C, N - q0, Qw, QW, lane, 64, 1.f);
    stage_rows(gb + (long)q0 * C, C, N - q0, Gw, QW, lane, 64, 1.f);
    float L[16], Dr[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int row = q0 + acc_row(i, h);
      L[i] = row < N ? lse[(long)bh * N + row] : INFINITY;
      Dr[i] = row < N ? D[(long)bh * N + row] : 0.f;
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = 0.f; dP[i] = 0.f; }
#pragma unroll
      for (int j = 0; j < 16; ++j) {
        const float2 qf = *reinterpret_cast<const float2*>(&Qw[r * ALD + 4 * j + 2 * h]);
        const float2 gf = *reinterpret_cast<const float2*>(&Gw[r * ALD + 4 * j + 2 * h]);
        const float2 kf = *reinterpret_cast<const float2*>(&Ks[(32 * t + r) * ALD + 4 * j + 2 * h]);
        const float2 vf = *reinterpret_cast<const float2*>(&Vs[(32 * t + r) * ALD + 4 * j + 2 * h]);
        S = mfma32(qf.x, kf.x, S);
        dP = mfma32(gf.x, vf.x, dP);
        S = mfma32(qf.y, kf.y, S);
        dP = mfma32(gf.y, vf.y, dP);
      }
      const int keyi = c * KC + 32 * t + r;
      const bool kok = keyi < Nk;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int rl = acc_row(i, h);
        const float p = kok ? expf(S[i] * scale - L[i]) : 0.f;
        float mk = 1.f;
        if (thresh != 0u) mk = fs_dropout_keep((uint32_t)(((long)bh * N + q0 + rl) * Nk) + (uint32_t)keyi, key, thresh) ? drop_scale : 0.f;
        const float pt = p * mk;
        const float ds = p * (mk * dP[i] - Dr[i]) * scale;
        // contract over the query (row) index: register i of this lane is A[key r][query rl]
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          dV[t][dt] = mfma32(pt, Gw[rl * ALD + 32 * dt + r], dV[t][dt]);
          dK[t][dt] = mfma32(ds, Qw[rl * ALD + 32 * dt + r], dK[t][dt]);
        }
      }
    }
    wave_sync();
  }
  // sum the four waves' partial tiles in wave order: red[tensor][key][d]
  __syncthreads();
  float* red = &Qs[0][0];                              // 2 * 64 * 64 floats = 32 KB <= 4 * QW * ALD floats
  for (int w = 0; w < 4; ++w) {
    if (wave == w) {
#pragma unroll
      for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int dt = 0; dt < 2; ++dt)
#pragma unroll
          for (int i = 0; i < 16; ++i) {
            const int a0 = (32 * t + acc_row(i, h)) * HD + 32 * dt + r;
            red[a0] = (w == 0 ? 0.f : red[a0]) + dK[t][dt][i];
            red[KC * HD + a0] = (w == 0 ? 0.f : red[KC * HD + a0]) + dV[t][dt][i];
          }
    }
    __syncthreads();
  }
  for (int i = tid; i < 2 * KC * 16; i += 256) {
    const int which = i / (KC * 16), rem = i - which * KC * 16;
    const int kk = rem >> 4, c4 = (rem & 15) * 4;
    const int keyi = c * KC + kk;
    if (keyi >= Nk) continue;
    const float* src = red + which * KC * HD + kk * HD + c4;
    float* dst = (which ? dv : dk) + ((long)b * Nk + keyi) * C + hd * HD + c4;
    if (atomics) {
      atomicAdd(dst, src[0]); atomicAdd(dst + 1, src[1]); atomicAdd(dst + 2, src[2]); atomicAdd(dst + 3, src[3]);
    } else {
      *reinterpret_cast<f32x4*>(dst) = f32x4{src[0], src[1], src[2], src[3]};
    }
  }
}

}  // namespace

extern "C" {

int fs_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long M, int C,
                     float eps, hipStream_t stream) {
  FS_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  const int need = cdiv(C, 64);
  const dim3 grid16((unsigned)cdiv(M, 16));
#define FS_LN_FWD16(NJ_) hipLaunchKernelGGL(layernorm_fwd16_kernel<NJ_>, grid16, dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, C, eps)
  if (need <= 1) FS_LN_FWD16(1);
  else if (need <= 2) FS_LN_FWD16(2);
  else if (need <= 5) FS_LN_FWD16(5);
  else hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, C, eps);
#undef FS_LN_FWD16
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// accumulate != 0: dgamma / dbeta are ADDED to (gradient-arena targets: no memset), else overwritten
// rows per workgroup / workgroups of the two layernorm backward kernels
static void ln_bwd_plan(long M, int C, int* rpb, int* nblk) {
  if (cdiv(C, 64) <= 5) {           // 16 rows in flight per workgroup; ~2048 workgroups (8 per CU)
    int r = (int)((M + 2047) / 2048); r = (r + 15) / 16 * 16; if (r < 16) r = 16;
    *rpb = r;
  } else {
    int r = (int)((M + 1023) / 1024); if (r < 4) r = 4;
    *rpb = r;
  }
  *nblk = cdiv(M, *rpb);
}
long fs_layernorm_bwd_scratch_floats(long M, int C) {
  if (M <= 0 || C <= 0) return 0;
  int rpb, nblk;
  ln_bwd_plan(M, C, &rpb, &nblk);
  return 2L * nblk * C;
}

// scratch: fs_layernorm_bwd_scratch_floats(M, C) floats -- per-workgroup records of the dgamma / dbeta sums, added in workgroup order
// addend (nullable, may alias nothing of the outputs): a second gradient of x, added to dx in the same pass (include/fovealseg.h: fs_layernorm_bwd_add)
int fs_layernorm_bwd_add(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, const float* addend, float* dx,
                         float* dgamma, float* dbeta, long M, int C, int accumulate, float* scratch, hipStream_t stream) {
  FS_REQUIRE(g && x && gamma && mean && rstd && dx && dgamma && dbeta && scratch && M > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  int rpb, nblk;
  ln_bwd_plan(M, C, &rpb, &nblk);
  float* part_g = scratch;
  float* part_b = scratch + (long)nblk * C;
  const int need = cdiv(C, 64);
  const size_t lds = 8 * (size_t)C * sizeof(float);
  if (need <= 5) {                  // (wider rows fill a wave per row; 8 float4s per lane of five arrays would spill)
    const dim3 grid((unsigned)nblk);
#define FS_LN_BWD16(NJ_) hipLaunchKernelGGL(layernorm_bwd16_kernel<NJ_>, grid, dim3(256), lds, stream, g, x, gamma, mean, rstd, dx, part_g, part_b, M, C, rpb, addend)
    if (need <= 1) FS_LN_BWD16(1);
    else if (need <= 2) FS_LN_BWD16(2);
    else FS_LN_BWD16(5);
#undef FS_LN_BWD16
  } else {
    hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(nblk), dim3(256), lds, stream, g, x, gamma, mean, rstd,
                       dx, part_g, part_b, M, C, rpb, addend);
  }
  FS_LAUNCH_CHECK();
  return fs_slab_reduce_pair(part_g, part_b, nblk, C, dgamma, dbeta, accumulate, stream);
}

int fs_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                     float* dgamma, float* dbeta, long M, int C, int accumulate, float* scratch, hipStream_t stream) {
  return fs_layernorm_bwd_add(g, x, gamma, mean, rstd, nullptr, dx, dgamma, dbeta, M, C, accumulate, scratch, stream);
}

int fs_gelu_fwd(const float* x, float* y, long n, hipStream_t stream) {
  FS_REQUIRE(x && y && n > 0 && n % 4 == 0);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n / 4, 1.f, 0u, 0u);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_gelu_bwd(const float* g, const float* x, float* dx, long n, hipStream_t stream) {
  FS_REQUIRE(g && x && dx && n > 0 && n % 4 == 0);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, g, x, dx, n / 4, 1.f, 0u, 0u);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
// y = dropout_p(gelu(x)) and its backward dx = g * mask / (1-p) * gelu'(x) in one pass each (the mask of fs_dropout with the same key on
// the same element index); n < 2^32
int fs_gelu_dropout_fwd(const float* x, float* y, long n, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(x && y && n > 0 && n % 4 == 0 && n < 4294967296L && drop_p > 0.f && drop_p < 1.f);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n / 4, 1.0f / (float)(1.0 - (double)drop_p),
                     (uint32_t)((double)drop_p * 4294967296.0), key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_gelu_dropout_bwd(const float* g, const float* x, float* dx, long n, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(g && x && dx && n > 0 && n % 4 == 0 && n < 4294967296L && drop_p > 0.f && drop_p < 1.f);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, g, x, dx, n / 4, 1.0f / (float)(1.0 - (double)drop_p),
                     (uint32_t)((double)drop_p * 4294967296.0), key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// flip = 0 forward (bias nullable), flip = 1 input gradient (pass bias = NULL)
int fs_dwconv3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int flip,
                   hipStream_t stream) {
  FS_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  const long items = (long)B * cdiv(H, DW_RY) * cdiv(W, DW_SEG);
  const long threads = items * (C / 4);
  FS_REQUIRE(threads < (1L << 31) * 256);
  hipLaunchKernelGGL(dwconv3_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, w, bias, y, B, H, W, C, flip);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// number of slab rows ([9][C] floats each) fs_dwconv3_bwd_weight needs as scratch for a (B, H, W, C) problem
int fs_dwconv3_wgrad_lanes(int B, int H, int W, int C) {
  if (B <= 0 || H <= 0 || W <= 0 || C <= 0 || C % 4) return 0;
  const long items = (long)B * cdiv(H, DW_RY) * cdiv(W, DW_SEG);
  long lanes = (256L * 256) / (C / 4);           // about one 256-thread block per CU in all
  if (lanes < 1) lanes = 1;
  if (lanes > items) lanes = items;
  return (int)lanes;
}

// accumulate != 0: dw is ADDED to (gradient-arena target), else overwritten.  ws = fs_dwconv3_wgrad_lanes() * 9 * C floats of scratch.
int fs_dwconv3_bwd_weight(const float* x, const float* dy, float* dw, float* ws, int B, int H, int W, int C, int accumulate,
                          hipStream_t stream) {
  FS_REQUIRE(x && dy && dw && ws && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  const int lanes = fs_dwconv3_wgrad_lanes(B, H, W, C);
  const long threads = (long)lanes * (C / 4);
  hipLaunchKernelGGL(dwconv3_wgrad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, dy, ws, B, H, W, C, lanes, 9);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(dwconv3_wgrad_reduce_kernel, dim3(cdiv(9L * C, 32)), dim3(256), 0, stream, ws, lanes, C, dw, accumulate, 9, nullptr, 0);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_dwconv3_bwd_weight_bias -- the same two launches also form db[C] = column sums of dy (ws = lanes * 10 * C floats)
int fs_dwconv3_bwd_weight_bias(const float* x, const float* dy, float* dw, float* db, float* ws, int B, int H, int W, int C, int accumulate_w,
                               int accumulate_b, hipStream_t stream) {
  FS_REQUIRE(x && dy && dw && db && ws && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  const int lanes = fs_dwconv3_wgrad_lanes(B, H, W, C);
  const long threads = (long)lanes * (C / 4);
  hipLaunchKernelGGL(dwconv3_wgrad_kernel, dim3((unsigned)((threads + 255) / 256)), dim3(256), 0, stream, x, dy, ws, B, H, W, C, lanes, 10);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(dwconv3_wgrad_reduce_kernel, dim3(cdiv(10L * C, 32)), dim3(256), 0, stream, ws, lanes, C, dw, accumulate_w, 10, db, accumulate_b);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// out = x + keep_b * y / (1-p); per_sample = elements per batch sample; x nullable (then out = scaled y: the backward)
int fs_residual_droppath(const float* x, const float* y, float* out, long n, long per_sample, float drop_p, uint32_t key,
                         hipStream_t stream) {
  FS_REQUIRE(y && out && n > 0 && n % 4 == 0 && per_sample > 0 && per_sample % 4 == 0 && drop_p >= 0.f && drop_p < 1.f);
  float scale = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { scale = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(residual_droppath_kernel, dim3(blocks), dim3(256), 0, stream, x, y, out, n / 4, per_sample / 4, scale, thresh,
                     key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_droppath_dropout_bwd
int fs_droppath_dropout_bwd(const float* g, float* dz, long n, long per_sample, float droppath_p, uint32_t droppath_key, float drop_p,
                            uint32_t drop_key, hipStream_t stream) {
  FS_REQUIRE(g && dz && n > 0 && n % 4 == 0 && n < 4294967296L && per_sample > 0 && per_sample % 4 == 0);
  FS_REQUIRE(droppath_p >= 0.f && droppath_p < 1.f && drop_p >= 0.f && drop_p < 1.f);
  float dps = 1.f, ds = 1.f; uint32_t dpt = 0u, dt = 0u;
  if (droppath_p > 0.f) { dps = 1.0f / (float)(1.0 - (double)droppath_p); dpt = (uint32_t)((double)droppath_p * 4294967296.0); }
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); dt = (uint32_t)((double)drop_p * 4294967296.0); }
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(droppath_dropout_bwd_kernel, dim3(blocks), dim3(256), 0, stream, g, dz, n / 4, per_sample / 4, dps, dpt, droppath_key, ds, dt, drop_key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// col[B*Ho*Wo][Kp] <- patches of x (B,H,W,C): element (r, s, c) of a k x k patch, zero where the patch leaves the image and in the padding
// columns k*k*C .. Kp-1.  Kp % 4 == 0, Kp >= k*k*C.
int fs_unfold(const float* x, float* col, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, hipStream_t stream) {
  FS_REQUIRE(x && col && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && Kp % 4 == 0 && Kp >= k * k * C);
  FS_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1);
  const long total4 = (long)B * Ho * Wo * (Kp / 4);
  long blocks = (total4 + 255) / 256; if (blocks > 65536) blocks = 65536;
  if (stride == k && pad == 0 && C % 4 == 0 && Kp == k * k * C) {
    hipLaunchKernelGGL(unfold_patch_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, col, H, W, C, k, Ho, Wo, total4);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  hipLaunchKernelGGL(unfold_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, x, col, B, H, W, C, k, stride, pad, Ho, Wo, Kp, total4);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
// the adjoint: dx (B,H,W,C) overwritten with the sum of every patch element that came from it
int fs_fold(const float* col, float* dx, int B, int H, int W, int C, int k, int stride, int pad, int Ho, int Wo, int Kp, hipStream_t stream) {
  FS_REQUIRE(col && dx && B > 0 && H > 0 && W > 0 && C > 0 && k > 0 && stride > 0 && pad >= 0 && Kp % 4 == 0 && Kp >= k * k * C);
  FS_REQUIRE(Ho == (H + 2 * pad - k) / stride + 1 && Wo == (W + 2 * pad - k) / stride + 1);
  const long total = (long)B * H * W * C;
  long blocks = (total + 255) / 256; if (blocks > 65536) blocks = 65536;
  if (stride == k && pad == 0 && C % 4 == 0 && Kp == k * k * C) {
    long b4 = (total / 4 + 255) / 256; if (b4 > 65536) b4 = 65536;
    hipLaunchKernelGGL(fold_patch_kernel, dim3((unsigned)b4), dim3(256), 0, stream, col, dx, H, W, C, k, Ho, Wo, total / 4);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  hipLaunchKernelGGL(fold_kernel, dim3((unsigned)blocks), dim3(256), 0, stream, col, dx, B, H, W, C, k, stride, pad, Ho, Wo, Kp, total);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int N, int Nk, int heads,
                     float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && o && lse && B > 0 && N > 0 && Nk > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f);
  FS_REQUIRE((long)B * heads * N * Nk < 4294967296L && (long)B * heads < 65536);
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  hipLaunchKernelGGL(attn_mfma_fwd_kernel, dim3(cdiv(N, 128), B * heads), dim3(256), 0, stream, q, k, v, o, lse, N, Nk, heads, scale, ds,
                     thresh, key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_attention_bwd_dq_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                              const unsigned* mask, float* dq, void* ws, long ws_bytes, int B, int N, int Nk, int heads, float scale,
                              float drop_p, uint32_t key, hipStream_t stream);       // attention_split.hip
int fs_attention_bwd_dkv_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                               const unsigned* mask, float* dk, float* dv, float* parts, int B, int N, int Nk, int heads, float scale,
                               float drop_p, uint32_t key, hipStream_t stream);
long fs_attention_bwd_split_ws_bytes(int B, int Nk, int heads);
long fs_attention_bwd_split_parts_offset(int B, int Nk, int heads);

static int attention_bwd_impl(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse, float* dq,
                              float* dk, float* dv, float* scratch, const unsigned* mask, void* ws, long ws_bytes, int B, int N, int Nk,
                              int heads, float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && o && go && lse && dq && dk && dv && scratch && B > 0 && N > 0 && Nk > 0 && heads > 0);
  FS_REQUIRE(drop_p >= 0.f && drop_p < 1.f && (long)B * heads * N * Nk < 4294967296L && (long)B * heads < 65536);
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  const long rows = (long)B * heads * N;
  hipLaunchKernelGGL(attn_rowdot_kernel, dim3(cdiv(rows * 16, 256)), dim3(256), 0, stream, go, o, scratch, B, N, heads);
  FS_LAUNCH_CHECK();
  if (ws != nullptr) {          // split precision: all three gradients by attention_split.hip
    const int e = fs_attention_bwd_dq_split(q, k, v, go, lse, scratch, mask, dq, ws, ws_bytes, B, N, Nk, heads, scale, drop_p, key, stream);
    if (e != FS_OK) return e;
    float* parts = ws_bytes >= fs_attention_bwd_split_ws_bytes(B, Nk, heads)
                       ? reinterpret_cast<float*>(static_cast<unsigned char*>(ws) + fs_attention_bwd_split_parts_offset(B, Nk, heads)) : nullptr;
    return fs_attention_bwd_dkv_split(q, k, v, go, lse, scratch, mask, dk, dv, parts, B, N, Nk, heads, scale, drop_p, key, stream);
  } else {
    hipLaunchKernelGGL(attn_mfma_bwd_dq_kernel, dim3(cdiv(N, 128), B * heads), dim3(256), 0, stream, q, k, v, go, lse, scratch, dq, N, Nk,
                       heads, scale, ds, thresh, key);
    FS_LAUNCH_CHECK();
  }
  // key chunks x (batch, head) workgroups; when that does not fill the chip the query range is split too (atomics into zeroed dk/dv)
  const int nkc = cdiv(Nk, KC), nqb = cdiv(N, 128);
  int nsplit = cdiv(512, (long)nkc * B * heads);
  if (nsplit > nqb) nsplit = nqb;
  if (nsplit > 64) nsplit = 64;
  if (fs_deterministic()) nsplit = 1;        // deterministic mode: one query range per key chunk, no atomics (the split-precision kernels sum partial tensors in order instead)
  const int bps = cdiv(nqb, nsplit);
  nsplit = cdiv(nqb, bps);
  if (nsplit > 1) {
    const size_t kvbytes = (size_t)B * Nk * heads * HD * sizeof(float);
    hipError_t e = hipMemsetAsync(dk, 0, kvbytes, stream);
    if (e != hipSuccess) return (int)e;
    e = hipMemsetAsync(dv, 0, kvbytes, stream);
    if (e != hipSuccess) return (int)e;
  }
  hipLaunchKernelGGL(attn_mfma_bwd_dkv_kernel, dim3(nkc, B * heads, nsplit), dim3(256), 0, stream, q, k, v, go, lse, scratch, dk, dv, N, Nk,
                     heads, scale, ds, thresh, key, bps, nsplit > 1 ? 1 : 0);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// dq / dk / dv overwritten.  o = the forward's output; scratch = B*heads*N floats (row sums of dO * O).
int fs_attention_bwd(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse, float* dq,
                     float* dk, float* dv, float* scratch, int B, int N, int Nk, int heads, float scale, float drop_p, uint32_t key,
                     hipStream_t stream) {
  return attention_bwd_impl(q, k, v, o, go, lse, dq, dk, dv, scratch, nullptr, nullptr, 0, B, N, Nk, heads, scale, drop_p, key, stream);
}

// the same with the split-precision (bf16x3) kernels; ws = fs_attention_bwd_split_ws_bytes(B, Nk, heads) bytes; mask = the keep words
// fs_attention_fwd_split left (nullable: the kernels then hash again)
int fs_attention_bwd_split(const float* q, const float* k, const float* v, const float* o, const float* go, const float* lse,
                           const unsigned* mask, float* dq, float* dk, float* dv, float* scratch, void* ws, long ws_bytes, int B, int N, int Nk,
                           int heads, float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(ws != nullptr);
  return attention_bwd_impl(q, k, v, o, go, lse, dq, dk, dv, scratch, mask, ws, ws_bytes, B, N, Nk, heads, scale, drop_p, key, stream);
}

}  // extern "C"
