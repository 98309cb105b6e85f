// SegFormer (Mix-Transformer) encoder pieces that are not convolutions / linears:
//   LayerNorm over channels, exact GELU, depthwise 3x3 conv (+bias), sequence-reduced multi-head attention
//   (<= 128 key/value tokens, head_dim 64), residual + DropPath.
// Replaces the ATen ops behind transformers==4.46.2 modeling_segformer (third-party; call sites
// models/segformer.py:2,9-11,33-37,88-100).  Tokens are rows of an NHWC tensor: (B, N=H*W, C).
#include "common.h"

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

// dx = rstd * (g*gamma - mean(g*gamma) - xhat * mean(g*gamma*xhat));  dgamma += g*xhat, dbeta += g (atomics per block)
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                            const float* __restrict__ gamma, const float* __restrict__ mean,
                                                            const float* __restrict__ rstd, float* __restrict__ dx,
                                                            float* __restrict__ dgamma, float* __restrict__ dbeta, long M, int C,
                                                            int rows_per_block) {
  extern __shared__ float sm[];          // [2][C] block partials of dgamma / dbeta
  for (int c = threadIdx.x; c < 2 * C; c += blockDim.x) sm[c] = 0.f;
  __syncthreads();
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
      if (c < C) *reinterpret_cast<f32x4*>(dx + row * C + c) = rs * (gg[j] - m1 - xh[j] * m2);
    }
  }
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < C) {
#pragma unroll
      for (int e = 0; e < 4; ++e) { atomicAdd(&sm[c + e], ag[j][e]); atomicAdd(&sm[C + c + e], ab[j][e]); }
    }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += blockDim.x) { atomicAdd(&dgamma[c], sm[c]); atomicAdd(&dbeta[c], sm[C + c]); }
}

// ------------------------------------------------------------------------------------------
// exact GELU
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gelu_fwd_kernel(const float* __restrict__ x, float* __restrict__ y, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) v[j] = 0.5f * v[j] * (1.f + erff(v[j] * 0.70710678118654752440f));
    reinterpret_cast<f32x4*>(y)[i] = v;
  }
}
__global__ __launch_bounds__(256) void gelu_bwd_kernel(const float* __restrict__ g, const float* __restrict__ x,
                                                       float* __restrict__ dx, long n4) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const f32x4 v = reinterpret_cast<const f32x4*>(x)[i];
    f32x4 d = reinterpret_cast<const f32x4*>(g)[i];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float cdf = 0.5f * (1.f + erff(v[j] * 0.70710678118654752440f));
      const float pdf = 0.39894228040143267794f * expf(-0.5f * v[j] * v[j]);
      d[j] *= cdf + v[j] * pdf;
    }
    reinterpret_cast<f32x4*>(dx)[i] = d;
  }
}

// ------------------------------------------------------------------------------------------
// depthwise 3x3 conv, stride 1, pad 1, NHWC; weights logical (C,1,3,3) = [c][9]
// flip = 0: y = sum_t w[c][t] x[p + t] (+ bias);  flip = 1: input gradient (taps mirrored, no bias)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void dwconv3_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                      const float* __restrict__ bias, float* __restrict__ y, int B, int H, int W,
                                                      int C, int flip) {
  const int cw = C >> 2;
  const long total = (long)B * H * W * cw;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = 4 * (int)(i % cw);
    const long pix = i / cw;
    const int px = (int)(pix % W), py = (int)((pix / W) % H), b = (int)(pix / ((long)W * H));
    f32x4 acc = bias ? *reinterpret_cast<const f32x4*>(bias + c) : f32x4{0, 0, 0, 0};
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int dy = t / 3 - 1, dxx = t % 3 - 1;
      const int iy = py + dy, ix = px + dxx;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      const int wt = flip ? 8 - t : t;
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + (((long)b * H + iy) * W + ix) * C + c);
      const f32x4 wv = {w[(c + 0) * 9 + wt], w[(c + 1) * 9 + wt], w[(c + 2) * 9 + wt], w[(c + 3) * 9 + wt]};
      acc += xv * wv;
    }
    *reinterpret_cast<f32x4*>(y + pix * C + c) = acc;
  }
}
// dw[c][t] = sum_pix x[pix + t][c] * dy[pix][c]; thread = (channel quad), block = chunk of pixels, atomics at the end
__global__ __launch_bounds__(1024) void dwconv3_wgrad_kernel(const float* __restrict__ x, const float* __restrict__ dy,
                                                            float* __restrict__ dw, int B, int H, int W, int C, int pix_per_block) {
  const int cw = C >> 2;
  const int cg = threadIdx.x % cw, pl = threadIdx.x / cw, npl = blockDim.x / cw;
  if (pl >= npl) return;
  const int c = 4 * cg;
  const long P = (long)B * H * W;
  const long p0 = (long)blockIdx.x * pix_per_block;
  long p1 = p0 + pix_per_block; if (p1 > P) p1 = P;
  f32x4 acc[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) acc[t] = f32x4{0, 0, 0, 0};
  for (long p = p0 + pl; p < p1; p += npl) {
    const int px = (int)(p % W), py = (int)((p / W) % H), b = (int)(p / ((long)W * H));
    const f32x4 g = *reinterpret_cast<const f32x4*>(dy + p * C + c);
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int iy = py + t / 3 - 1, ix = px + t % 3 - 1;
      if (iy < 0 || iy >= H || ix < 0 || ix >= W) continue;
      acc[t] += g * *reinterpret_cast<const f32x4*>(x + (((long)b * H + iy) * W + ix) * C + c);
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int e = 0; e < 4; ++e) atomicAdd(&dw[(c + e) * 9 + t], acc[t][e]);
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

// ------------------------------------------------------------------------------------------
// Attention with a short key/value sequence (Nk <= 128 tokens after sequence reduction), head_dim 64.
// One thread per query row; K and V of the (batch, head) live in LDS and are read as broadcasts.
//   S = q.K^T * scale; P = softmax(S); P~ = dropout(P); O = P~ V;  lse saved for the backward.
// q (B,N,heads*64), k/v (B,Nk,heads*64), o (B,N,heads*64)
// ------------------------------------------------------------------------------------------
constexpr int HD = 64, NKMAX = 128;

__global__ __launch_bounds__(128) void attn_fwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, float* __restrict__ o, float* __restrict__ lse,
                                                       int N, int Nk, int heads, float scale, float drop_scale, uint32_t thresh,
                                                       uint32_t key) {
  __shared__ float Ks[NKMAX * HD], Vs[NKMAX * HD];
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * HD;
  for (int i = threadIdx.x; i < Nk * (HD / 4); i += blockDim.x) {
    const int j = i / (HD / 4), d = 4 * (i % (HD / 4));
    *reinterpret_cast<f32x4*>(&Ks[j * HD + d]) = *reinterpret_cast<const f32x4*>(k + ((long)b * Nk + j) * C + h * HD + d);
    *reinterpret_cast<f32x4*>(&Vs[j * HD + d]) = *reinterpret_cast<const f32x4*>(v + ((long)b * Nk + j) * C + h * HD + d);
  }
  __syncthreads();
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  if (row >= N) return;
  float qr[HD];
#pragma unroll
  for (int d = 0; d < HD; d += 4) {
    const f32x4 t = *reinterpret_cast<const f32x4*>(q + ((long)b * N + row) * C + h * HD + d);
    qr[d] = t.x * scale; qr[d + 1] = t.y * scale; qr[d + 2] = t.z * scale; qr[d + 3] = t.w * scale;
  }
  // pass 1: max and sum of exp (online), pass 2: probabilities -> output
  float m = -INFINITY, l = 0.f;
  for (int j = 0; j < Nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += qr[d] * Ks[j * HD + d];
    const float mn = fmaxf(m, s);
    l = l * expf(m - mn) + expf(s - mn);
    m = mn;
  }
  const float L = m + logf(l);
  float acc[HD];
#pragma unroll
  for (int d = 0; d < HD; ++d) acc[d] = 0.f;
  const uint32_t ebase = (uint32_t)(((long)bh * N + row) * Nk);
  for (int j = 0; j < Nk; ++j) {
    float s = 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) s += qr[d] * Ks[j * HD + d];
    float p = expf(s - L);
    if (thresh != 0u) p = fs_dropout_keep(ebase + j, key, thresh) ? p * drop_scale : 0.f;
#pragma unroll
    for (int d = 0; d < HD; ++d) acc[d] += p * Vs[j * HD + d];
  }
  lse[(long)bh * N + row] = L;
#pragma unroll
  for (int d = 0; d < HD; d += 4)
    *reinterpret_cast<f32x4*>(o + ((long)b * N + row) * C + h * HD + d) = f32x4{acc[d], acc[d + 1], acc[d + 2], acc[d + 3]};
}

// backward: per row recompute P; dQ per row; dK/dV accumulated per block in LDS, then fp32 atomics.
__global__ __launch_bounds__(128) void attn_bwd_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                       const float* __restrict__ v, const float* __restrict__ go,
                                                       const float* __restrict__ lse, float* __restrict__ dq,
                                                       float* __restrict__ dk, float* __restrict__ dv, int N, int Nk, int heads,
                                                       float scale, float drop_scale, uint32_t thresh, uint32_t key) {
  extern __shared__ float sm[];
  float* Ks = sm;                       // [Nk][64]
  float* Vs = Ks + NKMAX * HD;          // [Nk][64]
  float* dKs = Vs + NKMAX * HD;         // [Nk][64] block accumulators
  float* dVs = dKs + NKMAX * HD;
  const int bh = blockIdx.y, b = bh / heads, h = bh - b * heads;
  const int C = heads * HD;
  for (int i = threadIdx.x; i < Nk * (HD / 4); i += blockDim.x) {
    const int j = i / (HD / 4), d = 4 * (i % (HD / 4));
    *reinterpret_cast<f32x4*>(&Ks[j * HD + d]) = *reinterpret_cast<const f32x4*>(k + ((long)b * Nk + j) * C + h * HD + d);
    *reinterpret_cast<f32x4*>(&Vs[j * HD + d]) = *reinterpret_cast<const f32x4*>(v + ((long)b * Nk + j) * C + h * HD + d);
    *reinterpret_cast<f32x4*>(&dKs[j * HD + d]) = f32x4{0, 0, 0, 0};
    *reinterpret_cast<f32x4*>(&dVs[j * HD + d]) = f32x4{0, 0, 0, 0};
  }
  __syncthreads();
  const int row = blockIdx.x * blockDim.x + threadIdx.x;
  const bool active = row < N;
  float qr[HD], gr[HD], dqr[HD];
  float L = 0.f;
#pragma unroll
  for (int d = 0; d < HD; ++d) { qr[d] = 0.f; gr[d] = 0.f; }     // idle rows must contribute exact zeros
  if (active) {
#pragma unroll
    for (int d = 0; d < HD; d += 4) {
      const f32x4 t = *reinterpret_cast<const f32x4*>(q + ((long)b * N + row) * C + h * HD + d);
      const f32x4 u = *reinterpret_cast<const f32x4*>(go + ((long)b * N + row) * C + h * HD + d);
      qr[d] = t.x; qr[d + 1] = t.y; qr[d + 2] = t.z; qr[d + 3] = t.w;
      gr[d] = u.x; gr[d + 1] = u.y; gr[d + 2] = u.z; gr[d + 3] = u.w;
    }
    L = lse[(long)bh * N + row];
  }
#pragma unroll
  for (int d = 0; d < HD; ++d) dqr[d] = 0.f;
  const uint32_t ebase = (uint32_t)(((long)bh * N + row) * Nk);
  // D = sum_j p_j * dp_j  (dp_j = mask_j * drop_scale * (go . V_j))
  float Dsum = 0.f;
  if (active)
    for (int j = 0; j < Nk; ++j) {
      float s = 0.f, gv = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s += qr[d] * Ks[j * HD + d]; gv += gr[d] * Vs[j * HD + d]; }
      const float p = expf(s * scale - L);
      const float mk = (thresh == 0u) ? 1.f : (fs_dropout_keep(ebase + j, key, thresh) ? drop_scale : 0.f);
      Dsum += p * mk * gv;
    }
  for (int j = 0; j < Nk; ++j) {
    float ds = 0.f, pt = 0.f;
    if (active) {
      float s = 0.f, gv = 0.f;
#pragma unroll
      for (int d = 0; d < HD; ++d) { s += qr[d] * Ks[j * HD + d]; gv += gr[d] * Vs[j * HD + d]; }
      const float p = expf(s * scale - L);
      const float mk = (thresh == 0u) ? 1.f : (fs_dropout_keep(ebase + j, key, thresh) ? drop_scale : 0.f);
      pt = p * mk;                                  // dropped-out probability (multiplies V)
      ds = p * (mk * gv - Dsum) * scale;            // d(q.k_j)
#pragma unroll
      for (int d = 0; d < HD; ++d) dqr[d] += ds * Ks[j * HD + d];
    }
    // block reduction of dK_j += ds * q, dV_j += pt * go over the rows of this block (wave shuffle, then LDS atomics)
#pragma unroll
    for (int d = 0; d < HD; ++d) {
      float a = wave_sum(ds * qr[d]);
      float c2 = wave_sum(pt * gr[d]);
      if ((threadIdx.x & 63) == 0) { atomicAdd(&dKs[j * HD + d], a); atomicAdd(&dVs[j * HD + d], c2); }
    }
  }
  if (active) {
#pragma unroll
    for (int d = 0; d < HD; d += 4)
      *reinterpret_cast<f32x4*>(dq + ((long)b * N + row) * C + h * HD + d) = f32x4{dqr[d], dqr[d + 1], dqr[d + 2], dqr[d + 3]};
  }
  __syncthreads();
  for (int i = threadIdx.x; i < Nk * HD; i += blockDim.x) {
    const int j = i / HD, d = i - j * HD;
    atomicAdd(&dk[((long)b * Nk + j) * C + h * HD + d], dKs[i]);
    atomicAdd(&dv[((long)b * Nk + j) * C + h * HD + d], dVs[i]);
  }
}

}  // namespace

extern "C" {

int fs_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, long M, int C,
                     float eps, hipStream_t stream) {
  FS_REQUIRE(x && gamma && beta && y && mean && rstd && M > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  hipLaunchKernelGGL(layernorm_fwd_kernel, dim3(cdiv(M, 4)), dim3(256), 0, stream, x, gamma, beta, y, mean, rstd, M, C, eps);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// dgamma / dbeta are overwritten
int fs_layernorm_bwd(const float* g, const float* x, const float* gamma, const float* mean, const float* rstd, float* dx,
                     float* dgamma, float* dbeta, long M, int C, hipStream_t stream) {
  FS_REQUIRE(g && x && gamma && mean && rstd && dx && dgamma && dbeta && M > 0 && C > 0 && C % 4 == 0 && C <= 2048);
  hipError_t e = hipMemsetAsync(dgamma, 0, C * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(dbeta, 0, C * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  int rpb = (int)((M + 1023) / 1024); if (rpb < 4) rpb = 4;
  hipLaunchKernelGGL(layernorm_bwd_kernel, dim3(cdiv(M, rpb)), dim3(256), 2 * C * sizeof(float), stream, g, x, gamma, mean, rstd,
                     dx, dgamma, dbeta, M, C, rpb);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_gelu_fwd(const float* x, float* y, long n, hipStream_t stream) {
  FS_REQUIRE(x && y && n > 0 && n % 4 == 0);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, y, n / 4);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_gelu_bwd(const float* g, const float* x, float* dx, long n, hipStream_t stream) {
  FS_REQUIRE(g && x && dx && n > 0 && n % 4 == 0);
  int blocks = cdiv(n / 4, 256); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(gelu_bwd_kernel, dim3(blocks), dim3(256), 0, stream, g, x, dx, n / 4);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// flip = 0 forward (bias nullable), flip = 1 input gradient (pass bias = NULL)
int fs_dwconv3_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int C, int flip,
                   hipStream_t stream) {
  FS_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0);
  int blocks = cdiv((long)B * H * W * (C / 4), 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(dwconv3_kernel, dim3(blocks), dim3(256), 0, stream, x, w, bias, y, B, H, W, C, flip);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_dwconv3_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int C, hipStream_t stream) {
  FS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && C > 0 && C % 4 == 0 && C / 4 <= 1024);
  hipError_t e = hipMemsetAsync(dw, 0, (size_t)C * 9 * sizeof(float), stream);
  if (e != hipSuccess) return (int)e;
  const long P = (long)B * H * W;
  const int cw = C / 4;
  const int threads = cw >= 256 ? cw : (256 / cw) * cw;         // whole channel rows per block, <= 1024
  int ppb = (int)((P + 511) / 512); if (ppb < 16) ppb = 16;
  hipLaunchKernelGGL(dwconv3_wgrad_kernel, dim3(cdiv(P, ppb)), dim3(threads), 0, stream, x, dy, dw, B, H, W, C, ppb);
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

int fs_attention_fwd(const float* q, const float* k, const float* v, float* o, float* lse, int B, int N, int Nk, int heads,
                     float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && o && lse && B > 0 && N > 0 && Nk > 0 && Nk <= NKMAX && heads > 0 && drop_p >= 0.f && drop_p < 1.f);
  FS_REQUIRE((long)B * heads * N * Nk < 4294967296L);
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  hipLaunchKernelGGL(attn_fwd_kernel, dim3(cdiv(N, 128), B * heads), dim3(128), 0, stream, q, k, v, o, lse, N, Nk, heads, scale, ds,
                     thresh, key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// dq overwritten; dk / dv overwritten (zeroed here, accumulated with atomics)
int fs_attention_bwd(const float* q, const float* k, const float* v, const float* go, const float* lse, float* dq, float* dk,
                     float* dv, int B, int N, int Nk, int heads, float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && go && lse && dq && dk && dv && B > 0 && N > 0 && Nk > 0 && Nk <= NKMAX && heads > 0);
  const size_t kvbytes = (size_t)B * Nk * heads * HD * sizeof(float);
  hipError_t e = hipMemsetAsync(dk, 0, kvbytes, stream);
  if (e != hipSuccess) return (int)e;
  e = hipMemsetAsync(dv, 0, kvbytes, stream);
  if (e != hipSuccess) return (int)e;
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  static bool attr_set = false;
  const int smem = 4 * NKMAX * HD * (int)sizeof(float);
  if (!attr_set) {
    e = hipFuncSetAttribute((const void*)attn_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) return (int)e;
    attr_set = true;
  }
  hipLaunchKernelGGL(attn_bwd_kernel, dim3(cdiv(N, 128), B * heads), dim3(128), smem, stream, q, k, v, go, lse, dq, dk, dv, N, Nk,
                     heads, scale, ds, thresh, key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // extern "C"
