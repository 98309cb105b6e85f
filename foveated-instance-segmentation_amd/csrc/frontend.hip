// Foveation front-end kernels (HBM / gather bound):
//   K1  gaze map + bilinear low-res RGB -> 5-channel NHWC saliency input   (models/models.py:684-705)
//   K3  ReLU -> 1x1 conv 24->1 -> softmax over HxW                          (models/models.py:369-372,715-723)
//   K4  replication pad + separable Gaussian accumulation -> sampling grid  (models/models.py:594-637,819-821)
//   K5  non-uniform bilinear grid_sample (image -> NHWC, label -> int64)    (models/models.py:880,909,951)
//   K6  grid_sample backward w.r.t. the grid (+ scatter-add w.r.t. input)
//   K11 area pooling of the full-resolution mask + min/max-normalised MSE   (models/models.py:730,889-898)
#include "common.h"

namespace {

// ------------------------------------------------------------------------------------------
// K1
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gaze_lowres_kernel(const float* __restrict__ x, const float* __restrict__ focus,
                                                          float* __restrict__ out, int B, int H, int W, int hs, int ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * hs * ws) return;
  const int ox = (int)(i % ws), oy = (int)((i / ws) % hs), b = (int)(i / ((long)ws * hs));
  // bilinear, align_corners=False (ATen upsample_bilinear2d)
  const float sy = (float)H / (float)hs, sx = (float)W / (float)ws;
  float fy = sy * ((float)oy + 0.5f) - 0.5f; if (fy < 0.f) fy = 0.f;
  float fx = sx * ((float)ox + 0.5f) - 0.5f; if (fx < 0.f) fx = 0.f;
  const int y0 = (int)fy, x0 = (int)fx;
  const int y1 = y0 + (y0 < H - 1 ? 1 : 0), x1 = x0 + (x0 < W - 1 ? 1 : 0);
  const float ly1 = fy - (float)y0, ly0 = 1.f - ly1, lx1 = fx - (float)x0, lx0 = 1.f - lx1;
  float* o = out + i * 5;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const float* p = x + ((long)b * 3 + c) * H * W;
    const float v00 = p[(long)y0 * W + x0], v01 = p[(long)y0 * W + x1];
    const float v10 = p[(long)y1 * W + x0], v11 = p[(long)y1 * W + x1];
    o[c] = ly0 * (lx0 * v00 + lx1 * v01) + ly1 * (lx0 * v10 + lx1 * v11);
  }
  // gaze map: ((sqrt((i-h)^2+(j-w)^2)) / sqrt(hs^2+ws^2))^2
  const float h = focus[2 * b] * (float)(hs - 1), w = focus[2 * b + 1] * (float)(ws - 1);
  const float dy = (float)oy - h, dx = (float)ox - w;
  const float dist = sqrtf(dy * dy + dx * dx);
  const float maxd = (float)sqrt((double)hs * hs + (double)ws * ws);
  const float r = dist / maxd;
  o[3] = r * r;
  o[4] = r * r;
}

// ------------------------------------------------------------------------------------------
// K3: one workgroup per image
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void compress_softmax_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                                    const float* __restrict__ bias, float* __restrict__ xs,
                                                                    int HW, int C) {
  extern __shared__ float logit[];
  __shared__ float red[16];
  const int b = blockIdx.x;
  const float* sb = s + (long)b * HW * C;
  float lmax = -INFINITY;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    float acc = 0.f;
    for (int c = 0; c < C; ++c) {
      const float v = sb[(long)p * C + c];
      acc += (v < 0.f ? 0.f : v) * w[c];
    }
    acc += bias[0];
    logit[p] = acc;
    lmax = fmaxf(lmax, acc);
  }
  lmax = wave_max(lmax);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = lmax;
  __syncthreads();
  float m = red[0];
  for (int i = 1; i < (int)(blockDim.x >> 6); ++i) m = fmaxf(m, red[i]);
  float lsum = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float e = expf(logit[p] - m);
    logit[p] = e;
    lsum += e;
  }
  const float tot = block_sum<float>(lsum, red);
  for (int p = threadIdx.x; p < HW; p += blockDim.x) xs[(long)b * HW + p] = logit[p] / tot;
}

// dlogit = xs * (g - sum(g*xs)); ds[p][c] = dlogit*w[c]*(s>0); dw[c] += sum dlogit*relu(s); db += sum dlogit
// Round 5: CS_SLICES workgroups per image (round 4: one -- 64 workgroups on a 256-CU chip walked 78 MB at 0.36 TB/s).  Every workgroup of
// an image forms the image's softmax dot product itself (6 400 products: cheaper than a launch), then owns a contiguous slice of the pixels;
// the slice is walked as float4 channel quads (768 threads = a multiple of the C/4 quads of a pixel, so a thread keeps its channel quad
// and four dw accumulators).  Each workgroup leaves one record of C + 1 partial sums; fs_slab_reduce adds the records in index order.
constexpr int CS_SLICES = 8;
constexpr int CS_THREADS = 768;
__global__ __launch_bounds__(CS_THREADS) void compress_softmax_bwd_kernel(const float* __restrict__ g, const float* __restrict__ xs,
                                                                          const float* __restrict__ s, const float* __restrict__ w,
                                                                          float* __restrict__ ds, float* __restrict__ part /* [B*S][C], then [B*S] */,
                                                                          int HW, int C) {
  __shared__ float red[16];
  __shared__ f32x4 dwq[CS_THREADS];
  const int b = blockIdx.x / CS_SLICES, sl = blockIdx.x - b * CS_SLICES;
  const float* gb = g + (long)b * HW;
  const float* xb = xs + (long)b * HW;
  float dot = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) dot += gb[p] * xb[p];
  dot = block_sum<float>(dot, red);
  const int Q = C >> 2;                                    // channel quads per pixel (blockDim.x % Q == 0: checked by the launcher)
  const int per = (HW + CS_SLICES - 1) / CS_SLICES;
  const int p0 = sl * per, p1 = p0 + per < HW ? p0 + per : HW;
  const int cq = threadIdx.x % Q;
  const f32x4 wq = *reinterpret_cast<const f32x4*>(w + 4 * cq);
  const f32x4* s4 = reinterpret_cast<const f32x4*>(s + (long)b * HW * C);
  f32x4* ds4 = reinterpret_cast<f32x4*>(ds + (long)b * HW * C);
  f32x4 dwl = {0.f, 0.f, 0.f, 0.f};
  float dbl = 0.f;
  for (long e = (long)p0 * Q + threadIdx.x; e < (long)p1 * Q; e += blockDim.x) {
    const int p = (int)(e / Q);
    const float dl = xb[p] * (gb[p] - dot);
    const f32x4 v = s4[e];
    f32x4 o;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      o[j] = v[j] > 0.f ? dl * wq[j] : 0.f;
      dwl[j] += dl * (v[j] < 0.f ? 0.f : v[j]);
    }
    ds4[e] = o;
    if (cq == 0) dbl += dl;
  }
  dwq[threadIdx.x] = dwl;
  dbl = block_sum<float>(dbl, red);                        // (its barriers also publish dwq)
  if (threadIdx.x < C) {
    const int q = threadIdx.x >> 2, j = threadIdx.x & 3;
    float a = 0.f;
    for (int t = q; t < (int)blockDim.x; t += Q) a += dwq[t][j];      // threads with this channel quad, in thread order
    part[(long)blockIdx.x * C + threadIdx.x] = a;
  }
  if (threadIdx.x == 0) part[(long)gridDim.x * C + blockIdx.x] = dbl;
}

// CompressNet.forward on its own (models/models.py:360-372): logit[p] = w . relu(s[p]) + bias, one thread per pixel (C <= 32: the
// pixel's channels are 1-2 cache lines); backward: ds = g * w * (s > 0), dw += sum g * relu(s), db += sum g (one workgroup per image).
__global__ __launch_bounds__(256) void compress_fwd_kernel(const float* __restrict__ s, const float* __restrict__ w,
                                                           const float* __restrict__ bias, float* __restrict__ out, long n, int C) {
  const long p = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (p >= n) return;
  float acc = 0.f;
  for (int c = 0; c < C; ++c) {
    const float v = s[p * C + c];
    acc += (v < 0.f ? 0.f : v) * w[c];
  }
  out[p] = acc + bias[0];
}

__global__ __launch_bounds__(1024) void compress_bwd_kernel(const float* __restrict__ g, const float* __restrict__ s,
                                                            const float* __restrict__ w, float* __restrict__ ds,
                                                            float* __restrict__ part /* [B][C], then [B] */, int HW, int C) {
  __shared__ float red[16];
  __shared__ float dwacc[16 * 32];
  const int b = blockIdx.x;
  const float* gb = g + (long)b * HW;
  const float* sb = s + (long)b * HW * C;
  float* dsb = ds + (long)b * HW * C;
  float dbl = 0.f;
  float dwl[32];
#pragma unroll
  for (int c = 0; c < 32; ++c) dwl[c] = 0.f;
  for (int p = threadIdx.x; p < HW; p += blockDim.x) {
    const float dl = gb[p];
    dbl += dl;
#pragma unroll
    for (int c = 0; c < 32; ++c) {
      if (c < C) {
        const float v = sb[(long)p * C + c];
        dsb[(long)p * C + c] = v > 0.f ? dl * w[c] : 0.f;
        dwl[c] += dl * (v < 0.f ? 0.f : v);
      }
    }
  }
  // per-wave sums into their own LDS row, rows added in wave order; the image's record goes to part[b], summed in image order by
  // fs_slab_reduce (no atomics anywhere: the result does not depend on which wave or workgroup finishes first)
#pragma unroll
  for (int c = 0; c < 32; ++c) {
    if (c < C) {
      const float v = wave_sum(dwl[c]);
      if ((threadIdx.x & 63) == 0) dwacc[(threadIdx.x >> 6) * 32 + c] = v;
    }
  }
  dbl = block_sum<float>(dbl, red);
  __syncthreads();
  if (threadIdx.x < C) {
    float a = 0.f;
    for (int wv = 0; wv < (int)(blockDim.x >> 6); ++wv) a += dwacc[wv * 32 + threadIdx.x];
    part[(long)b * C + threadIdx.x] = a;
  }
  if (threadIdx.x == 0) part[(long)gridDim.x * C + b] = dbl;
}

// ------------------------------------------------------------------------------------------
// K11: area pool (adaptive average, windows [floor(o*H/h), ceil((o+1)*H/h)) ), one block per (b, oy)
// ------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void area_pool_kernel(const float* __restrict__ y, float* __restrict__ out, int H, int W,
                                                        int hs, int ws) {
  extern __shared__ float colsum[];   // W floats
  const int oy = blockIdx.x % hs, b = blockIdx.x / hs;
  const int y0 = (int)(((long)oy * H) / hs);
  const int y1 = (int)((((long)(oy + 1)) * H + hs - 1) / hs);
  const float* base = y + (long)b * H * W;
  if ((W & 3) == 0 && ((size_t)base & 15) == 0) {
    // 16-byte loads, the 13-14 rows of the window issued back to back (independent): the label plane is the only full-resolution
    // tensor the path reads, so this pass is pure HBM streaming.  Same row order per column as the scalar loop: bit-identical.
    for (int xq = threadIdx.x; xq < (W >> 2); xq += blockDim.x) {
      f32x4 s = {0.f, 0.f, 0.f, 0.f};
      int r = y0;
      for (; r + 3 < y1; r += 4) {
        const f32x4 a0 = *reinterpret_cast<const f32x4*>(base + (long)r * W + 4 * xq);
        const f32x4 a1 = *reinterpret_cast<const f32x4*>(base + (long)(r + 1) * W + 4 * xq);
        const f32x4 a2 = *reinterpret_cast<const f32x4*>(base + (long)(r + 2) * W + 4 * xq);
        const f32x4 a3 = *reinterpret_cast<const f32x4*>(base + (long)(r + 3) * W + 4 * xq);
        s += a0; s += a1; s += a2; s += a3;
      }
      for (; r < y1; ++r) s += *reinterpret_cast<const f32x4*>(base + (long)r * W + 4 * xq);
      *reinterpret_cast<f32x4*>(&colsum[4 * xq]) = s;
    }
  } else {
    for (int xcol = threadIdx.x; xcol < W; xcol += blockDim.x) {
      float s = 0.f;
      for (int r = y0; r < y1; ++r) s += base[(long)r * W + xcol];
      colsum[xcol] = s;
    }
  }
  __syncthreads();
  for (int ox = threadIdx.x; ox < ws; ox += blockDim.x) {
    const int x0 = (int)(((long)ox * W) / ws);
    const int x1 = (int)((((long)(ox + 1)) * W + ws - 1) / ws);
    float s = 0.f;
    for (int c = x0; c < x1; ++c) s += colsum[c];
    out[((long)b * hs + oy) * ws + ox] = s / (float)((y1 - y0) * (x1 - x0));
  }
}

// edge loss = coef * mean(((xs-min)/(max-min) - (t-tmin)/(tmax-tmin))^2), whole-batch min/max.
// stats out: [xs_min, xs_max, t_min, t_max, n_argmin, n_argmax] (+ scratch behind them, fs_edge_loss_stats_floats).
// Round 5: every pass over the batch is spread over EL_WGS(n) workgroups (round 4: ONE workgroup walked the 3.3 MB twice, 150 us forward
// and 259 us backward on a 256-CU chip).  A whole-batch statistic is two launches -- per-workgroup partials, then every workgroup of the
// next pass adds ALL partials itself in index order (<= 256 records: cheaper than a launch, and no result depends on workgroup timing):
//   forward : [min/max partials] -> [min/max, squared-error + arg-count partials] -> [sum, loss + stats]        (3 launches)
//   backward: [sum g (u - 1), sum g u partials] -> [sums, dxs]                                                   (2 launches)
// scratch layout (floats, behind the 6 stats + 2 pad): mm[G][4] | sums[G][4] doubles (acc, cmin, cmax, -) | bsum[G][2] doubles
constexpr int EL_THREADS = 256;
static int el_wgs(long n) {
  const long w = (n / 4 + EL_THREADS * 4 - 1) / (EL_THREADS * 4);          // >= 4 float4 per thread
  return (int)(w < 1 ? 1 : (w > 256 ? 256 : w));
}
struct ElPtrs { float* mm; double* sums; double* bsum; };
__host__ __device__ __forceinline__ ElPtrs el_ptrs(float* stats, int G) {
  ElPtrs p;
  p.mm = stats + 8;
  p.sums = reinterpret_cast<double*>(stats + 8 + 4 * G);
  p.bsum = p.sums + 4 * G;
  return p;
}

__global__ __launch_bounds__(EL_THREADS) void edge_minmax_kernel(const float* __restrict__ xs, const float* __restrict__ t, long n, float* __restrict__ stats) {
  __shared__ float red[4][4];
  const int G = gridDim.x;
  float mn = INFINITY, mx = -INFINITY, tmn = INFINITY, tmx = -INFINITY;
  const long n4 = ((((size_t)xs | (size_t)t) & 15) == 0) ? n / 4 : 0;
  const f32x4* xs4 = reinterpret_cast<const f32x4*>(xs);
  const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
  for (long i = (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n4; i += (long)G * EL_THREADS) {
    const f32x4 a = xs4[i], b = t4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) { mn = fminf(mn, a[e]); mx = fmaxf(mx, a[e]); tmn = fminf(tmn, b[e]); tmx = fmaxf(tmx, b[e]); }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n; i += (long)G * EL_THREADS) {
    const float a = xs[i], b = t[i];
    mn = fminf(mn, a); mx = fmaxf(mx, a); tmn = fminf(tmn, b); tmx = fmaxf(tmx, b);
  }
  mn = wave_min(mn); mx = wave_max(mx); tmn = wave_min(tmn); tmx = wave_max(tmx);
  const int wv = threadIdx.x >> 6;
  if ((threadIdx.x & 63) == 0) { red[0][wv] = mn; red[1][wv] = mx; red[2][wv] = tmn; red[3][wv] = tmx; }
  __syncthreads();
  if (threadIdx.x == 0) {
    float* mm = el_ptrs(stats, G).mm + 4 * blockIdx.x;
    mm[0] = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
    mm[1] = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
    mm[2] = fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3]));
    mm[3] = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
  }
}

// whole-batch min / max from the G partial records (min / max are order-independent); every thread ends up with all four
__device__ __forceinline__ void edge_global_minmax(const float* __restrict__ mm, int G, float (*red)[4], float& mn, float& mx, float& tmn, float& tmx) {
  mn = INFINITY; mx = -INFINITY; tmn = INFINITY; tmx = -INFINITY;
  for (int i = threadIdx.x; i < G; i += EL_THREADS) {
    mn = fminf(mn, mm[4 * i]); mx = fmaxf(mx, mm[4 * i + 1]); tmn = fminf(tmn, mm[4 * i + 2]); tmx = fmaxf(tmx, mm[4 * i + 3]);
  }
  mn = wave_min(mn); mx = wave_max(mx); tmn = wave_min(tmn); tmx = wave_max(tmx);
  const int wv = threadIdx.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) { red[0][wv] = mn; red[1][wv] = mx; red[2][wv] = tmn; red[3][wv] = tmx; }
  __syncthreads();
  mn = fminf(fminf(red[0][0], red[0][1]), fminf(red[0][2], red[0][3]));
  mx = fmaxf(fmaxf(red[1][0], red[1][1]), fmaxf(red[1][2], red[1][3]));
  tmn = fminf(fminf(red[2][0], red[2][1]), fminf(red[2][2], red[2][3]));
  tmx = fmaxf(fmaxf(red[3][0], red[3][1]), fmaxf(red[3][2], red[3][3]));
}

__global__ __launch_bounds__(EL_THREADS) void edge_sums_kernel(const float* __restrict__ xs, const float* __restrict__ t, long n, float* __restrict__ stats) {
  __shared__ float red[4][4];
  __shared__ double dred[16];
  const int G = gridDim.x;
  const ElPtrs P = el_ptrs(stats, G);
  float mn, mx, tmn, tmx;
  edge_global_minmax(P.mm, G, red, mn, mx, tmn, tmx);
  const float r = mx - mn, tr = tmx - tmn;
  double acc = 0.0, cmin = 0.0, cmax = 0.0;
  const long n4 = ((((size_t)xs | (size_t)t) & 15) == 0) ? n / 4 : 0;
  const f32x4* xs4 = reinterpret_cast<const f32x4*>(xs);
  const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
  for (long i = (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n4; i += (long)G * EL_THREADS) {
    const f32x4 a4 = xs4[i], b4 = t4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = (a4[e] - mn) / r, v = (b4[e] - tmn) / tr;
      const float d = u - v;
      acc += (double)(d * d);
      cmin += (a4[e] == mn) ? 1.0 : 0.0;
      cmax += (a4[e] == mx) ? 1.0 : 0.0;
    }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n; i += (long)G * EL_THREADS) {
    const float a = xs[i];
    const float u = (a - mn) / r, v = (t[i] - tmn) / tr;
    const float d = u - v;
    acc += (double)(d * d);
    cmin += (a == mn) ? 1.0 : 0.0;
    cmax += (a == mx) ? 1.0 : 0.0;
  }
  acc = block_sum<double>(acc, dred);
  cmin = block_sum<double>(cmin, dred);
  cmax = block_sum<double>(cmax, dred);
  if (threadIdx.x == 0) {
    double* o = P.sums + 4 * blockIdx.x;
    o[0] = acc; o[1] = cmin; o[2] = cmax; o[3] = 0.0;
    if (blockIdx.x == 0) { stats[0] = mn; stats[1] = mx; stats[2] = tmn; stats[3] = tmx; }
  }
}

__global__ __launch_bounds__(64) void edge_finish_kernel(long n, float coef, int G, float* __restrict__ loss, float* __restrict__ stats) {
  if (threadIdx.x != 0) return;
  const double* sums = el_ptrs(stats, G).sums;
  double acc = 0.0, cmin = 0.0, cmax = 0.0;
  for (int i = 0; i < G; ++i) { acc += sums[4 * i]; cmin += sums[4 * i + 1]; cmax += sums[4 * i + 2]; }      // index order
  loss[0] = coef * (float)(acc / (double)n);
  stats[4] = (float)cmin; stats[5] = (float)cmax;
}

__global__ __launch_bounds__(EL_THREADS) void edge_bwd_sums_kernel(const float* __restrict__ xs, const float* __restrict__ t, long n, float coef,
                                                                  const float* __restrict__ gout, float* __restrict__ stats) {
  __shared__ double dred[16];
  const int G = gridDim.x;
  const float mn = stats[0], mx = stats[1], tmn = stats[2], tmx = stats[3];
  const float r = mx - mn, tr = tmx - tmn;
  const float k = 2.f * coef * gout[0] / (float)n;
  double s_gu1 = 0.0, s_gu = 0.0;    // sum g*(u-1), sum g*u
  const long n4 = ((((size_t)xs | (size_t)t) & 15) == 0) ? n / 4 : 0;
  const f32x4* xs4 = reinterpret_cast<const f32x4*>(xs);
  const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
  for (long i = (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n4; i += (long)G * EL_THREADS) {
    const f32x4 a4 = xs4[i], b4 = t4[i];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = (a4[e] - mn) / r, v = (b4[e] - tmn) / tr;
      const float g = k * (u - v);
      s_gu1 += (double)(g * (u - 1.f));
      s_gu += (double)(g * u);
    }
  }
  for (long i = 4 * n4 + (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n; i += (long)G * EL_THREADS) {
    const float u = (xs[i] - mn) / r, v = (t[i] - tmn) / tr;
    const float g = k * (u - v);
    s_gu1 += (double)(g * (u - 1.f));
    s_gu += (double)(g * u);
  }
  s_gu1 = block_sum<double>(s_gu1, dred);
  s_gu = block_sum<double>(s_gu, dred);
  if (threadIdx.x == 0) {
    double* o = el_ptrs(stats, G).bsum + 2 * blockIdx.x;
    o[0] = s_gu1; o[1] = s_gu;
  }
}

__global__ __launch_bounds__(EL_THREADS) void edge_bwd_apply_kernel(const float* __restrict__ xs, const float* __restrict__ t, long n, float coef,
                                                                   const float* __restrict__ gout, const float* __restrict__ stats_c,
                                                                   float* __restrict__ dxs) {
  __shared__ double tot[2];
  const int G = gridDim.x;
  const float mn = stats_c[0], mx = stats_c[1], tmn = stats_c[2], tmx = stats_c[3];
  const float r = mx - mn, tr = tmx - tmn;
  const float k = 2.f * coef * gout[0] / (float)n;
  if (threadIdx.x == 0) {
    const double* bs = el_ptrs(const_cast<float*>(stats_c), G).bsum;
    double a = 0.0, b = 0.0;
    for (int i = 0; i < G; ++i) { a += bs[2 * i]; b += bs[2 * i + 1]; }      // index order
    tot[0] = a; tot[1] = b;
  }
  __syncthreads();
  const float dmn = (float)(tot[0] / (double)r) / stats_c[4];
  const float dmx = (float)(-tot[1] / (double)r) / stats_c[5];
  const long n4 = ((((size_t)xs | (size_t)t | (size_t)dxs) & 15) == 0) ? n / 4 : 0;
  const f32x4* xs4 = reinterpret_cast<const f32x4*>(xs);
  const f32x4* t4 = reinterpret_cast<const f32x4*>(t);
  for (long i = (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n4; i += (long)G * EL_THREADS) {
    const f32x4 a4 = xs4[i], b4 = t4[i];
    f32x4 d4;
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float u = (a4[e] - mn) / r, v = (b4[e] - tmn) / tr;
      float d = k * (u - v) / r;
      if (a4[e] == mn) d += dmn;
      if (a4[e] == mx) d += dmx;
      d4[e] = d;
    }
    reinterpret_cast<f32x4*>(dxs)[i] = d4;
  }
  for (long i = 4 * n4 + (long)blockIdx.x * EL_THREADS + threadIdx.x; i < n; i += (long)G * EL_THREADS) {
    const float a = xs[i];
    const float u = (a - mn) / r, v = (t[i] - tmn) / tr;
    float d = k * (u - v) / r;
    if (a == mn) d += dmn;
    if (a == mx) d += dmx;
    dxs[i] = d;
  }
}

// ------------------------------------------------------------------------------------------
// K4: Gaussian-saliency accumulation -> grid.  One workgroup per image, everything in LDS.
//   p  = sum_{r,s} g[r]g[s] x~[oy+r][ox+s]
//   ax = sum g[r]g[s] x~ * cx(ox+s),  cx(j) = (j-pad)/(ws-1)     ay likewise with cy(i)=(i-pad)/(hs-1)
//   gx = clamp(2ax/p-1), gy = clamp(2ay/p-1); x~ = replication-padded saliency.
// The filter is separable (makeGaussian, models/models.py:157), so it runs as a row pass (91 taps)
// and a column pass (91 taps) with double accumulators instead of an 8281-tap direct conv.
// ------------------------------------------------------------------------------------------
constexpr int GMAX = 6400;   // hs*ws upper bound of the LDS layout (80x80)
// Round 5: GG_BANDS workgroups per image, each owning a band of grid COLUMNS (round 4: one workgroup per image -- 64 workgroups on a
// 256-CU chip, 258 us for the backward, bound by the LDS reads of its 91-tap passes).  A row pass needs every column of its source row
// but produces only the band's columns; a column pass needs every row of the band's columns only -- so forward (row pass, column pass)
// is one launch of B x GG_BANDS workgroups with nothing exchanged, and backward (row, column, transposed row, transposed column) is two:
// the transposed row pass reads the (dp, dax, day) of ALL columns, which the first launch leaves in a scratch of 3 floats per grid point.
constexpr int GG_BANDS = 4;

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// TRAIN.def_saliency_pad_mode (models/models.py:819-825): which source index the padded position t = j - pad reads.
//   PAD_REPLICATION  nn.ReplicationPad2d: the border pixel;   PAD_REFLECT  F.pad(mode='reflect'): mirrored about the border pixel
//   (torch asks pad <= n - 1);   PAD_ZERO  F.pad(mode='constant'): nothing (-1).  The mode is a template parameter, so the default
//   (replication) instantiation is the code it was before the other two existed.
constexpr int PAD_REPLICATION = 0, PAD_REFLECT = 1, PAD_ZERO = 2;
template <int MODE>
__device__ __forceinline__ int padmap(int t, int n) {
  if (MODE == PAD_REPLICATION) return clampi(t, 0, n - 1);
  if (MODE == PAD_REFLECT) return t < 0 ? -t : (t > n - 1 ? 2 * (n - 1) - t : t);
  return (t < 0 || t > n - 1) ? -1 : t;
}

// row pass on the band [x0, x0 + bw): R0[y][ox - x0] = sum_s g[s] xs[y][padmap(ox+s-pad)],  R1 = same * cx(ox+s)
template <int MODE>
__device__ void gauss_rows(const float* xs, const double* g, int hs, int ws, int pad, int x0, int bw, float* R0, float* R1) {
  const int K = 2 * pad + 1;
  const double inv = 1.0 / (double)(ws - 1);
  for (int i = threadIdx.x; i < hs * bw; i += blockDim.x) {
    const int y = i / bw, ox = x0 + (i - y * bw);
    double a0 = 0.0, a1 = 0.0;
    for (int s = 0; s < K; ++s) {
      const int j = ox + s;
      const int sx = padmap<MODE>(j - pad, ws);
      if (MODE == PAD_ZERO && sx < 0) continue;
      const double v = g[s] * (double)xs[y * ws + sx];
      a0 += v;
      a1 += v * ((double)(j - pad) * inv);
    }
    R0[i] = (float)a0; R1[i] = (float)a1;
  }
}

// column pass at (oy, band column xb): R0 / R1 are [hs][bw]
template <int MODE>
__device__ __forceinline__ void gauss_cols_at(const float* R0, const float* R1, const double* g, int hs, int bw, int pad,
                                              int oy, int xb, double& p, double& ax, double& ay) {
  const int K = 2 * pad + 1;
  const double inv = 1.0 / (double)(hs - 1);
  p = 0.0; ax = 0.0; ay = 0.0;
  for (int r = 0; r < K; ++r) {
    const int i = oy + r;
    const int y = padmap<MODE>(i - pad, hs);
    if (MODE == PAD_ZERO && y < 0) continue;
    const double v0 = g[r] * (double)R0[y * bw + xb];
    p += v0;
    ax += g[r] * (double)R1[y * bw + xb];
    ay += v0 * ((double)(i - pad) * inv);
  }
}

__device__ __forceinline__ void gg_band(int ws, int& x0, int& bw, int& b) {
  b = blockIdx.x / GG_BANDS;
  const int band = blockIdx.x - b * GG_BANDS;
  const int w = (ws + GG_BANDS - 1) / GG_BANDS;
  x0 = band * w;
  bw = x0 >= ws ? 0 : (x0 + w <= ws ? w : ws - x0);
}

template <int MODE>
__global__ __launch_bounds__(1024) void gauss_grid_fwd_kernel(const float* __restrict__ xs_g, const double* __restrict__ g1d,
                                                              float* __restrict__ grid, int hs, int ws, int pad) {
  extern __shared__ float sm[];
  float* X = sm; float* R0 = sm + GMAX; float* R1 = sm + 2 * GMAX;       // (R0 / R1 hold the band only)
  __shared__ double g[256];
  int x0, bw, b;
  gg_band(ws, x0, bw, b);
  if (bw == 0) return;                                                   // workgroup-uniform
  const int n = hs * ws;
  for (int i = threadIdx.x; i < 2 * pad + 1; i += blockDim.x) g[i] = g1d[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) X[i] = xs_g[(long)b * n + i];
  __syncthreads();
  gauss_rows<MODE>(X, g, hs, ws, pad, x0, bw, R0, R1);
  __syncthreads();
  for (int i = threadIdx.x; i < hs * bw; i += blockDim.x) {
    const int oy = i / bw, xb = i - oy * bw;
    double p, ax, ay;
    gauss_cols_at<MODE>(R0, R1, g, hs, bw, pad, oy, xb, p, ax, ay);
    float gx = (float)(ax / p * 2.0 - 1.0), gy = (float)(ay / p * 2.0 - 1.0);
    gx = fminf(fmaxf(gx, -1.f), 1.f);
    gy = fminf(fmaxf(gy, -1.f), 1.f);
    const long o = ((long)b * n + oy * ws + x0 + xb) * 2;
    grid[o + 0] = gx;
    grid[o + 1] = gy;
  }
}

// backward, launch 1 of 2: dgrid (B,hs,ws,2) -> (dp, dax, day) per grid point of the band -> scratch [B][3][n]
template <int MODE>
__global__ __launch_bounds__(1024) void gauss_grid_bwd1_kernel(const float* __restrict__ xs_g, const double* __restrict__ g1d,
                                                               const float* __restrict__ dgrid, float* __restrict__ scratch,
                                                               int hs, int ws, int pad) {
  extern __shared__ float sm[];
  float* X = sm; float* R0 = sm + GMAX; float* R1 = sm + 2 * GMAX;
  __shared__ double g[256];
  int x0, bw, b;
  gg_band(ws, x0, bw, b);
  if (bw == 0) return;
  const int n = hs * ws;
  for (int i = threadIdx.x; i < 2 * pad + 1; i += blockDim.x) g[i] = g1d[i];
  for (int i = threadIdx.x; i < n; i += blockDim.x) X[i] = xs_g[(long)b * n + i];
  __syncthreads();
  gauss_rows<MODE>(X, g, hs, ws, pad, x0, bw, R0, R1);
  __syncthreads();
  float* S = scratch + (long)b * 3 * n;
  for (int i = threadIdx.x; i < hs * bw; i += blockDim.x) {
    const int oy = i / bw, xb = i - oy * bw;
    const int pt = oy * ws + x0 + xb;
    double p, ax, ay;
    gauss_cols_at<MODE>(R0, R1, g, hs, bw, pad, oy, xb, p, ax, ay);
    const double ux = ax / p * 2.0 - 1.0, uy = ay / p * 2.0 - 1.0;
    // clamp(-1,1) passes the gradient where the un-clamped value lies inside [-1,1] (bounds included)
    const double dgx = (ux >= -1.0 && ux <= 1.0) ? (double)dgrid[((long)b * n + pt) * 2 + 0] : 0.0;
    const double dgy = (uy >= -1.0 && uy <= 1.0) ? (double)dgrid[((long)b * n + pt) * 2 + 1] : 0.0;
    const double dax = 2.0 * dgx / p, day = 2.0 * dgy / p;
    const double dp = -(dax * ax + day * ay) / p;
    S[pt] = (float)dp; S[n + pt] = (float)dax; S[2 * n + pt] = (float)day;
  }
}

// padded positions j in [0, n + 2 pad) that read source index x under reflect (the pixel itself, its mirror image in the low pad when
// 1 <= x <= pad, its mirror image in the high pad when n-1-pad <= x <= n-2) or zero padding (the pixel itself): at most three
template <int MODE>
__device__ __forceinline__ int pad_sources(int x, int n, int pad, int* js) {
  int c = 0;
  js[c++] = x + pad;
  if (MODE == PAD_REFLECT) {
    if (x >= 1 && x <= pad) js[c++] = pad - x;
    if (x <= n - 2 && x >= n - 1 - pad) js[c++] = pad + 2 * (n - 1) - x;
  }
  return c;
}

// backward, launch 2 of 2: the transposed separable filter on (dp, dax, day), the padding folded back onto the pixels it copies (replication:
// the border pixels collect pad + 1 positions each; reflect: up to three positions per pixel; zero: one); the band's columns of dxs.
template <int MODE>
__global__ __launch_bounds__(1024) void gauss_grid_bwd2_kernel(const float* __restrict__ scratch, const double* __restrict__ g1d,
                                                               float* __restrict__ dxs, int hs, int ws, int pad) {
  extern __shared__ float sm[];
  float* A = sm; float* D = sm + GMAX; float* E = sm + 2 * GMAX; float* Bf = sm + 3 * GMAX; float* Cf = sm + 4 * GMAX;    // Bf / Cf: [hs][bw]
  __shared__ double g[256];
  constexpr int BT = 128;                              // largest grid side the border tables cover
  __shared__ double gx_lo[2 * BT], gx_hi[2 * BT], gy_lo[2 * BT], gy_hi[2 * BT];
  int x0, bw, b;
  gg_band(ws, x0, bw, b);
  if (bw == 0) return;
  const int n = hs * ws, K = 2 * pad + 1;
  for (int i = threadIdx.x; i < K; i += blockDim.x) g[i] = g1d[i];
  const float* S = scratch + (long)b * 3 * n;
  for (int i = threadIdx.x; i < n; i += blockDim.x) { A[i] = S[i]; D[i] = S[n + i]; E[i] = S[2 * n + i]; }
  __syncthreads();
  // The two border columns (rows) collect the pad+1 padded positions that replicate them.  Summed tap by tap that is
  // (pad+1) x K products for 2/ws of the pixels -- the threads that own them run ~pad times longer than the rest -- so the
  // sums over the padded positions are folded into per-source weights first:
  //   T0[o] = sum_j g[j-o],  T1[o] = sum_j c(j) g[j-o]   over the border's j range (left/top: [0,pad], right/bottom: [n-1+pad, n+2pad-1])
  const double invx = 1.0 / (double)(ws - 1), invy = 1.0 / (double)(hs - 1);
  const bool tables = MODE == PAD_REPLICATION && hs <= BT && ws <= BT;
  if (tables) {
    for (int q = threadIdx.x; q < 2 * (ws + hs); q += blockDim.x) {
      const bool isx = q < 2 * ws;
      const int qq = isx ? q : q - 2 * ws;
      const int len = isx ? ws : hs;
      const bool hi = qq >= len;
      const int o = hi ? qq - len : qq;
      const double inv = isx ? invx : invy;
      const int j_lo = hi ? len - 1 + pad : 0, j_hi = hi ? len + 2 * pad - 1 : pad;
      double t0 = 0.0, t1 = 0.0;
      for (int j = j_lo; j <= j_hi; ++j) {
        const int k = j - o;
        if (k < 0 || k >= K) continue;
        t0 += g[k];
        t1 += (double)(j - pad) * inv * g[k];
      }
      double* T = isx ? (hi ? gx_hi : gx_lo) : (hi ? gy_hi : gy_lo);
      T[2 * o] = t0; T[2 * o + 1] = t1;
    }
  }
  __syncthreads();
  // row pass (transposed, folded): for every source row oy and folded column x of the band:
  //   Va[oy][x] = sum_{j -> x} sum_ox g[j-ox] (D0 + cx(j) D1)[oy][ox],   Vb[oy][x] = sum_{j->x} sum_ox g[j-ox] D2[oy][ox]
  for (int i = threadIdx.x; i < hs * bw; i += blockDim.x) {
    const int oy = i / bw, x = x0 + (i - oy * bw);
    if (tables && (x == 0 || x == ws - 1)) {
      const double* T = x == 0 ? gx_lo : gx_hi;
      double va = 0.0, vb = 0.0;
      for (int ox = 0; ox < ws; ++ox) {
        va += T[2 * ox] * (double)A[oy * ws + ox] + T[2 * ox + 1] * (double)D[oy * ws + ox];
        vb += T[2 * ox] * (double)E[oy * ws + ox];
      }
      Bf[i] = (float)va; Cf[i] = (float)vb;
      continue;
    }
    const int j_lo = (x == 0) ? 0 : x + pad, j_hi = (x == ws - 1) ? ws + 2 * pad - 1 : x + pad;
    int js[3];
    const int nj = MODE == PAD_REPLICATION ? j_hi - j_lo + 1 : pad_sources<MODE>(x, ws, pad, js);
    double va = 0.0, vb = 0.0;
    for (int q = 0; q < nj; ++q) {
      const int j = MODE == PAD_REPLICATION ? j_lo + q : js[q];
      const double cx = (double)(j - pad) * invx;
      int o_lo = j - (K - 1); if (o_lo < 0) o_lo = 0;
      int o_hi = j; if (o_hi > ws - 1) o_hi = ws - 1;
      double s0 = 0.0, s1 = 0.0, s2 = 0.0;
      for (int ox = o_lo; ox <= o_hi; ++ox) {
        const double gv = g[j - ox];
        s0 += gv * (double)A[oy * ws + ox];
        s1 += gv * (double)D[oy * ws + ox];
        s2 += gv * (double)E[oy * ws + ox];
      }
      va += s0 + cx * s1;
      vb += s2;
    }
    Bf[i] = (float)va; Cf[i] = (float)vb;
  }
  __syncthreads();
  // column pass (transposed, folded): dxs[y][x] = sum_{i -> y} sum_oy g[i-oy] (Va + cy(i) Vb)[oy][x]
  for (int idx = threadIdx.x; idx < hs * bw; idx += blockDim.x) {
    const int y = idx / bw, xb = idx - y * bw;
    const long o = (long)b * n + y * ws + x0 + xb;
    if (tables && (y == 0 || y == hs - 1)) {
      const double* T = y == 0 ? gy_lo : gy_hi;
      double acc = 0.0;
      for (int oy = 0; oy < hs; ++oy) acc += T[2 * oy] * (double)Bf[oy * bw + xb] + T[2 * oy + 1] * (double)Cf[oy * bw + xb];
      dxs[o] = (float)acc;
      continue;
    }
    const int i_lo = (y == 0) ? 0 : y + pad, i_hi = (y == hs - 1) ? hs + 2 * pad - 1 : y + pad;
    int is[3];
    const int ni = MODE == PAD_REPLICATION ? i_hi - i_lo + 1 : pad_sources<MODE>(y, hs, pad, is);
    double acc = 0.0;
    for (int q = 0; q < ni; ++q) {
      const int i = MODE == PAD_REPLICATION ? i_lo + q : is[q];
      const double cy = (double)(i - pad) * invy;
      int o_lo = i - (K - 1); if (o_lo < 0) o_lo = 0;
      int o_hi = i; if (o_hi > hs - 1) o_hi = hs - 1;
      double s0 = 0.0, s1 = 0.0;
      for (int oy = o_lo; oy <= o_hi; ++oy) {
        const double gv = g[i - oy];
        s0 += gv * (double)Bf[oy * bw + xb];
        s1 += gv * (double)Cf[oy * bw + xb];
      }
      acc += s0 + cy * s1;
    }
    dxs[o] = (float)acc;
  }
}

// ------------------------------------------------------------------------------------------
// K5 / K6: grid_sample, bilinear, zeros padding, align_corners=False.  Arithmetic follows the
// bit-exact recipe of SURVEY.md §8(a)-A12 (verified against ATen's CPU kernel by the goldens):
//   ix = fma(gx+1, W/2, -0.5); w = ix-floor(ix); weights nw=(1-n)(1-w) ...; acc = nw*v_nw;
//   acc = fma(v_ne,ne,acc); acc = fma(v_sw,sw,acc); acc = fma(v_se,se,acc).
// Explicit __f*_rn intrinsics keep hipcc from re-associating or contracting differently.
// ------------------------------------------------------------------------------------------
struct Taps {
  int x0, y0;
  float nw, ne, sw, se;
  bool okx0, okx1, oky0, oky1;
  float w, n, e, s;   // fractional parts (east/south weights) and their complements
};
__device__ __forceinline__ Taps make_taps(float gx, float gy, int H, int W) {
  Taps t;
  const float ix = __fmaf_rn(__fadd_rn(gx, 1.f), (float)W * 0.5f, -0.5f);
  const float iy = __fmaf_rn(__fadd_rn(gy, 1.f), (float)H * 0.5f, -0.5f);
  const float fx = floorf(ix), fy = floorf(iy);
  t.w = __fsub_rn(ix, fx); t.e = __fsub_rn(1.f, t.w);
  t.n = __fsub_rn(iy, fy); t.s = __fsub_rn(1.f, t.n);
  t.nw = __fmul_rn(t.s, t.e); t.ne = __fmul_rn(t.s, t.w);
  t.sw = __fmul_rn(t.n, t.e); t.se = __fmul_rn(t.n, t.w);
  // floor of a possibly huge/NaN coordinate: clamp before the int conversion
  const float cx = fminf(fmaxf(fx, -2.f), (float)W + 1.f), cy = fminf(fmaxf(fy, -2.f), (float)H + 1.f);
  t.x0 = (int)cx; t.y0 = (int)cy;
  t.okx0 = (t.x0 >= 0) & (t.x0 < W); t.okx1 = (t.x0 + 1 >= 0) & (t.x0 + 1 < W);
  t.oky0 = (t.y0 >= 0) & (t.y0 < H); t.oky1 = (t.y0 + 1 >= 0) & (t.y0 + 1 < H);
  return t;
}
__device__ __forceinline__ float sample_plane(const float* __restrict__ p, int W, const Taps& t) {
  const float vnw = (t.oky0 & t.okx0) ? p[(long)t.y0 * W + t.x0] : 0.f;
  const float vne = (t.oky0 & t.okx1) ? p[(long)t.y0 * W + t.x0 + 1] : 0.f;
  const float vsw = (t.oky1 & t.okx0) ? p[(long)(t.y0 + 1) * W + t.x0] : 0.f;
  const float vse = (t.oky1 & t.okx1) ? p[(long)(t.y0 + 1) * W + t.x0 + 1] : 0.f;
  float acc = __fmul_rn(vnw, t.nw);
  acc = __fmaf_rn(vne, t.ne, acc);
  acc = __fmaf_rn(vsw, t.sw, acc);
  acc = __fmaf_rn(vse, t.se, acc);
  return acc;
}

// x (B,C,H,W) NCHW -> out (B,h,w,C) NHWC  [nhwc_out=1]  or (B,C,h,w) NCHW [nhwc_out=0]
__global__ __launch_bounds__(256) void grid_sample_fwd_kernel(const float* __restrict__ x, const float* __restrict__ grid,
                                                              float* __restrict__ out, int B, int C, int H, int W, int h,
                                                              int w, int nhwc_out) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * h * w) return;
  const int b = (int)(i / ((long)h * w));
  const long pix = i - (long)b * h * w;
  const Taps t = make_taps(grid[2 * i], grid[2 * i + 1], H, W);
  for (int c = 0; c < C; ++c) {
    const float v = sample_plane(x + ((long)b * C + c) * H * W, W, t);
    if (nhwc_out) out[i * C + c] = v;
    else out[((long)b * C + c) * h * w + pix] = v;
  }
}

// y (B,1,H,W) float mask -> label (B,h,w) int64 = trunc(bilinear(y))   (models/models.py:880,951)
__global__ __launch_bounds__(256) void grid_sample_label_kernel(const float* __restrict__ y, const float* __restrict__ grid,
                                                                long long* __restrict__ label, float* __restrict__ ysamp,
                                                                int B, int H, int W, int h, int w) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * h * w) return;
  const int b = (int)(i / ((long)h * w));
  const Taps t = make_taps(grid[2 * i], grid[2 * i + 1], H, W);
  const float v = sample_plane(y + (long)b * H * W, W, t);
  if (ysamp != nullptr) ysamp[i] = v;
  label[i] = (long long)v;
}

// gout (B,h,w,C) NHWC [nhwc=1] or (B,C,h,w): dgrid (B,h,w,2)
__global__ __launch_bounds__(256) void grid_sample_bwd_grid_kernel(const float* __restrict__ gout, const float* __restrict__ x,
                                                                   const float* __restrict__ grid, float* __restrict__ dgrid,
                                                                   int B, int C, int H, int W, int h, int w, int nhwc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * h * w) return;
  const int b = (int)(i / ((long)h * w));
  const long pix = i - (long)b * h * w;
  const Taps t = make_taps(grid[2 * i], grid[2 * i + 1], H, W);
  float gix = 0.f, giy = 0.f;
  for (int c = 0; c < C; ++c) {
    const float* p = x + ((long)b * C + c) * H * W;
    const float go = nhwc ? gout[i * C + c] : gout[((long)b * C + c) * h * w + pix];
    const float vnw = (t.oky0 & t.okx0) ? p[(long)t.y0 * W + t.x0] : 0.f;
    const float vne = (t.oky0 & t.okx1) ? p[(long)t.y0 * W + t.x0 + 1] : 0.f;
    const float vsw = (t.oky1 & t.okx0) ? p[(long)(t.y0 + 1) * W + t.x0] : 0.f;
    const float vse = (t.oky1 & t.okx1) ? p[(long)(t.y0 + 1) * W + t.x0 + 1] : 0.f;
    gix += go * ((vne - vnw) * t.s + (vse - vsw) * t.n);
    giy += go * ((vsw - vnw) * t.e + (vse - vne) * t.w);
  }
  dgrid[2 * i] = gix * ((float)W * 0.5f);
  dgrid[2 * i + 1] = giy * ((float)H * 0.5f);
}

// scatter-add w.r.t. the input image (dx zeroed by the launcher); not on the default path
// (x needs no gradient) -- provided for loss_at_high_res / the inverse warp.
__global__ __launch_bounds__(256) void grid_sample_bwd_input_kernel(const float* __restrict__ gout, const float* __restrict__ grid,
                                                                    float* __restrict__ dx, int B, int C, int H, int W, int h,
                                                                    int w, int nhwc) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * h * w) return;
  const int b = (int)(i / ((long)h * w));
  const long pix = i - (long)b * h * w;
  const Taps t = make_taps(grid[2 * i], grid[2 * i + 1], H, W);
  for (int c = 0; c < C; ++c) {
    float* p = dx + ((long)b * C + c) * H * W;
    const float go = nhwc ? gout[i * C + c] : gout[((long)b * C + c) * h * w + pix];
    if (t.oky0 & t.okx0) atomicAdd(&p[(long)t.y0 * W + t.x0], go * t.nw);
    if (t.oky0 & t.okx1) atomicAdd(&p[(long)t.y0 * W + t.x0 + 1], go * t.ne);
    if (t.oky1 & t.okx0) atomicAdd(&p[(long)(t.y0 + 1) * W + t.x0], go * t.sw);
    if (t.oky1 & t.okx1) atomicAdd(&p[(long)(t.y0 + 1) * W + t.x0 + 1], go * t.se);
  }
}

// integer index maps of the inverse deformation (models/models.py:644-645): trunc toward zero
__global__ void inverse_index_kernel(const float* __restrict__ grid, long long* __restrict__ u, long long* __restrict__ v,
                                     long n, int H, int W) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float gx = grid[2 * i], gy = grid[2 * i + 1];
  const float fu = __fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(W - 1));
  const float fv = __fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(H - 1));
  u[i] = (long long)(int)fu;
  v[i] = (long long)(int)fv;
}

// ---- inverse (un-foveating) warp, SURVEY §8(f)-3 -------------------------------------------------------------------
// models/models.py:639-655: every grid point i = (yi, xi) of the (h,w) sampling grid claims the full-resolution pixel
// (v,u) it was sampled from; duplicate claims resolve as ATen-CPU index_put_ does (the LAST index wins = largest i).
__global__ void inverse_owner_kernel(const float* __restrict__ grid, int* __restrict__ owner, int B, int hw, int Hs, int Ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * hw) return;
  const int b = (int)(i / hw), p = (int)(i - (long)b * hw);
  const float gx = grid[2 * i], gy = grid[2 * i + 1];
  const int u = (int)__fmul_rn(__fmul_rn(__fadd_rn(gx, 1.f), 0.5f), (float)(Ws - 1));
  const int v = (int)__fmul_rn(__fmul_rn(__fadd_rn(gy, 1.f), 0.5f), (float)(Hs - 1));
  if (u < 0 || u >= Ws || v < 0 || v >= Hs) return;
  atomicMax(&owner[((long)b * Hs + v) * Ws + u], p);
}
// grid_inv[b,v,u] = (xi/w*2-1, yi/h*2-1) of the owning grid point, 0 where nobody claims the pixel (the reference writes NaN
// and replaces it by 0 before sampling, models.py:931-932; the hole mask is owner < 0).
__global__ void inverse_grid_kernel(const int* __restrict__ owner, float* __restrict__ inv, long n, int h, int w) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const int p = owner[i];
  float gx = 0.f, gy = 0.f;
  if (p >= 0) {
    const int yi = p / w, xi = p - yi * w;
    gx = __fsub_rn(__fmul_rn(__fdiv_rn((float)xi, (float)w), 2.f), 1.f);
    gy = __fsub_rn(__fmul_rn(__fdiv_rn((float)yi, (float)h), 2.f), 1.f);
  }
  inv[2 * i] = gx; inv[2 * i + 1] = gy;
}
// Nearest-valid fill (models.py:159-286 with rev_deform_interp='nearest'): exact Euclidean nearest claimed pixel, ties to
// the smallest (row, col).  Pass A: nearest claimed column in the same row; pass B: minimise (y-y')^2 + (x-x'(y'))^2 over rows.
__global__ __launch_bounds__(256) void fill_row_nearest_kernel(const int* __restrict__ owner, int* __restrict__ rowx, int Ws) {
  extern __shared__ int rowbuf[];                      // [Ws]: claimed flag, then nearest claimed column
  __shared__ int carryL[256], carryR[256];
  const int tid = threadIdx.x;
  const long row = blockIdx.x;
  const int* o = owner + row * Ws;
  int* r = rowx + row * Ws;
  int any = 0;
  for (int x = tid; x < Ws; x += 256) { const int c = o[x] >= 0; rowbuf[x] = c; any |= c; }
  // most rows of a strongly magnified image hold no claimed pixel at all
  if (!__syncthreads_or(any)) {
    for (int x = tid; x < Ws; x += 256) r[x] = -1;
    return;
  }
  // thread t owns the columns [x0, x1): last / first claimed column of the segment, then a scan over the 256 segments
  const int seg = (Ws + 255) / 256;
  const int x0 = tid * seg < Ws ? tid * seg : Ws, x1 = x0 + seg < Ws ? x0 + seg : Ws;
  int last = -1, first = 0x7fffffff;
  for (int x = x0; x < x1; ++x)
    if (rowbuf[x]) { last = x; if (first == 0x7fffffff) first = x; }
  carryL[tid] = last; carryR[tid] = first;
  __syncthreads();
  for (int off = 1; off < 256; off <<= 1) {
    int l = carryL[tid], rr = carryR[tid];
    if (tid >= off) { const int v = carryL[tid - off]; l = v > l ? v : l; }
    if (tid + off < 256) { const int v = carryR[tid + off]; rr = v < rr ? v : rr; }
    __syncthreads();
    carryL[tid] = l; carryR[tid] = rr;
    __syncthreads();
  }
  int left = tid > 0 ? carryL[tid - 1] : -1;
  int right = tid < 255 ? carryR[tid + 1] : 0x7fffffff;
  for (int x = x0; x < x1; ++x) {                      // nearest claimed column at or left of x
    if (rowbuf[x]) left = x;
    rowbuf[x] = left;
  }
  for (int x = x1 - 1; x >= x0; --x) {                 // ... against the nearest at or right of x; a tie goes to the left one
    const int l = rowbuf[x];
    if (l == x) right = x;
    int best;
    if (l < 0) best = right == 0x7fffffff ? -1 : right;
    else if (right == 0x7fffffff) best = l;
    else best = (x - l) <= (right - x) ? l : right;
    rowbuf[x] = best;
  }
  __syncthreads();
  for (int x = tid; x < Ws; x += 256) r[x] = rowbuf[x];
}
__global__ void fill_col_nearest_kernel(const int* __restrict__ rowx, int* __restrict__ src, long n, int Hs, int Ws) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const long per = (long)Hs * Ws;
  const long b = i / per;
  const long rem = i - b * per;
  const int y = (int)(rem / Ws), x = (int)(rem - (long)y * Ws);
  const int* rx = rowx + b * per;
  long best = -1;
  int bsrc = -1;
  for (int d = 0; d < Hs; ++d) {                       // rows by increasing |y - y'|: stop once dy^2 alone exceeds the best
    if (best >= 0 && (long)d * d > best) break;
    for (int sgn = 0; sgn < 2; ++sgn) {
      const int yy = sgn == 0 ? y - d : y + d;
      if (yy < 0 || yy >= Hs || (d == 0 && sgn == 1)) continue;
      const int xx = rx[(long)yy * Ws + x];
      if (xx < 0) continue;
      const long dd = (long)d * d + (long)(x - xx) * (x - xx);
      const int cand = yy * Ws + xx;
      if (best < 0 || dd < best || (dd == best && cand < bsrc)) { best = dd; bsrc = cand; }
    }
  }
  src[i] = bsrc;
}
__global__ void fill_copy_kernel(float* __restrict__ vals, const int* __restrict__ owner, const int* __restrict__ src, int C, long per,
                                 long n) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;       // over (b, pixel): the source index is looked up once for all classes
  if (i >= n) return;
  if (owner[i] >= 0) return;                           // claimed pixels keep their sampled value
  const int sp = src[i];
  if (sp < 0) return;                                  // image without any claimed pixel
  const long b = i / per;
  const long pix = i - b * per;
  float* v = vals + b * C * per;
  int c = 0;
  for (; c + 4 <= C; c += 4) {                          // four independent gathers in flight per trip
    const float a0 = v[(c + 0) * per + sp], a1 = v[(c + 1) * per + sp], a2 = v[(c + 2) * per + sp], a3 = v[(c + 3) * per + sp];
    v[(c + 0) * per + pix] = a0; v[(c + 1) * per + pix] = a1; v[(c + 2) * per + pix] = a2; v[(c + 3) * per + pix] = a3;
  }
  for (; c < C; ++c) v[c * per + pix] = v[c * per + sp];
}

}  // namespace

// ---- input pipeline: decoded uint8 sample -> padded float planes of the batch, on the device ----
// img (H,W,Ci) uint8 HWC (PIL 'RGBA' / 'RGB' memory order) -> X[b] (Cx,HP,WP) = ToTensor (u8 / 255) + F.pad(zeros);
// mask (H,W) uint8 -> Y[b] (1,HP,WP) = float + F.pad.  DynamicFocus/e_preprocess_scripts/dataset.py:127-142.
__global__ __launch_bounds__(256) void ingest_sample_kernel(const unsigned char* __restrict__ img, const unsigned char* __restrict__ mask,
                                                            float* __restrict__ X, float* __restrict__ Y, int H, int W, int Ci, int Cx,
                                                            int HP, int WP, int pad_left, int pad_top) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)HP * WP) return;
  const int y = (int)(idx / WP), x = (int)(idx - (long)y * WP);
  const int sy = y - pad_top, sx = x - pad_left;
  const bool in = sy >= 0 && sy < H && sx >= 0 && sx < W;
  const unsigned char* px = img + ((long)sy * W + sx) * Ci;
  for (int c = 0; c < Cx; ++c) X[(long)c * HP * WP + idx] = in ? (float)px[c] / 255.0f : 0.f;
  if (Y != nullptr) Y[idx] = in ? (float)mask[(long)sy * W + sx] : 0.f;
}

// ---- grid up-sampling: nn.Upsample(size=task_input_size, mode='bilinear') of the (B,2,hs,ws) deformation grid when the task
// network runs at a higher resolution than the saliency map (models/models.py:621-631; BASELINE configs[3]: 80x80 -> 160x160).
// grid (B,h,w,2) -> out (B,H,W,2), ATen upsample_bilinear2d, align_corners=False.
struct GLerp { int i0, i1; float l0, l1; };
__device__ __forceinline__ GLerp glerp(int d, int in, int out) {
  GLerp L;
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
__global__ __launch_bounds__(256) void grid_upsample_fwd_kernel(const float2* __restrict__ grid, float2* __restrict__ out, int B, int h, int w,
                                                                int H, int W) {
  const long total = (long)B * H * W;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int ox = (int)(i % W), oy = (int)((i / W) % H), b = (int)(i / ((long)W * H));
    const GLerp Ly = glerp(oy, h, H), Lx = glerp(ox, w, W);
    const float2* base = grid + (long)b * h * w;
    const float2 v00 = base[(long)Ly.i0 * w + Lx.i0], v01 = base[(long)Ly.i0 * w + Lx.i1];
    const float2 v10 = base[(long)Ly.i1 * w + Lx.i0], v11 = base[(long)Ly.i1 * w + Lx.i1];
    float2 r;
    r.x = Ly.l0 * (Lx.l0 * v00.x + Lx.l1 * v01.x) + Ly.l1 * (Lx.l0 * v10.x + Lx.l1 * v11.x);
    r.y = Ly.l0 * (Lx.l0 * v00.y + Lx.l1 * v01.y) + Ly.l1 * (Lx.l0 * v10.y + Lx.l1 * v11.y);
    out[i] = r;
  }
}
// transpose as a gather (integer factors): every source point sums the output pixels whose 2x2 footprint touches it
__global__ __launch_bounds__(256) void grid_upsample_bwd_kernel(const float2* __restrict__ g, float2* __restrict__ dgrid, int B, int h, int w,
                                                                int H, int W) {
  const long total = (long)B * h * w;
  const int fy = H / h, fx = W / w;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
    const int qx = (int)(i % w), qy = (int)((i / w) % h), b = (int)(i / ((long)w * h));
    int y_lo = fy * qy - fy / 2, y_hi = fy * qy + (3 * fy) / 2 - 1;
    int x_lo = fx * qx - fx / 2, x_hi = fx * qx + (3 * fx) / 2 - 1;
    if (y_lo < 0) y_lo = 0;
    if (x_lo < 0) x_lo = 0;
    if (y_hi > H - 1 || qy == h - 1) y_hi = H - 1;
    if (x_hi > W - 1 || qx == w - 1) x_hi = W - 1;
    float2 acc = {0.f, 0.f};
    for (int oy = y_lo; oy <= y_hi; ++oy) {
      const GLerp Ly = glerp(oy, h, H);
      const float wy = (Ly.i0 == qy ? Ly.l0 : 0.f) + (Ly.i1 == qy ? Ly.l1 : 0.f);
      if (wy == 0.f) continue;
      for (int ox = x_lo; ox <= x_hi; ++ox) {
        const GLerp Lx = glerp(ox, w, W);
        const float wx = (Lx.i0 == qx ? Lx.l0 : 0.f) + (Lx.i1 == qx ? Lx.l1 : 0.f);
        if (wx == 0.f) continue;
        const float2 v = g[((long)b * H + oy) * W + ox];
        acc.x += wy * wx * v.x; acc.y += wy * wx * v.y;
      }
    }
    dgrid[i] = acc;
  }
}

extern "C" {

int fs_ingest_sample(const unsigned char* img, const unsigned char* mask, float* X, float* Y, int b, int H, int W, int Ci, int Cx,
                     int pad_left, int pad_right, int pad_top, int pad_bottom, hipStream_t stream) {
  FS_REQUIRE(img && X && b >= 0 && H > 0 && W > 0 && Ci >= 1 && Cx >= 1 && Cx <= Ci && (Y == nullptr || mask != nullptr));
  FS_REQUIRE(pad_left >= 0 && pad_right >= 0 && pad_top >= 0 && pad_bottom >= 0);
  const int HP = H + pad_top + pad_bottom, WP = W + pad_left + pad_right;
  const long plane = (long)HP * WP;
  hipLaunchKernelGGL(ingest_sample_kernel, dim3(cdiv(plane, 256)), dim3(256), 0, stream, img, mask, X + (long)b * Cx * plane,
                     Y ? Y + (long)b * plane : nullptr, H, W, Ci, Cx, HP, WP, pad_left, pad_top);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_gaze_lowres_fwd(const float* x, const float* focus, float* out, int B, int H, int W, int hs, int ws,
                       hipStream_t stream) {
  FS_REQUIRE(x && focus && out && B > 0 && H > 0 && W > 0 && hs > 1 && ws > 1);
  hipLaunchKernelGGL(gaze_lowres_kernel, dim3(cdiv((long)B * hs * ws, 256)), dim3(256), 0, stream, x, focus, out, B, H, W, hs, ws);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_compress_fwd(const float* s, const float* w, const float* bias, float* out, int B, int HW, int C, hipStream_t stream) {
  FS_REQUIRE(s && w && bias && out && B > 0 && HW > 0 && C > 0 && C <= 32);
  const long n = (long)B * HW;
  hipLaunchKernelGGL(compress_fwd_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, s, w, bias, out, n, C);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// scratch: B * (C + 1) floats (per-image partial sums of dw and db, added in image order)
int fs_compress_bwd(const float* g, const float* s, const float* w, float* ds, float* dw, float* db, int B, int HW, int C,
                    float* scratch, hipStream_t stream) {
  FS_REQUIRE(g && s && w && ds && dw && db && scratch && B > 0 && HW > 0 && C > 0 && C <= 32);
  hipLaunchKernelGGL(compress_bwd_kernel, dim3(B), dim3(1024), 0, stream, g, s, w, ds, scratch, HW, C);
  FS_LAUNCH_CHECK();
  const int r = fs_slab_reduce(scratch, B, C, dw, 0, stream);
  if (r != FS_OK) return r;
  return fs_slab_reduce(scratch + (long)B * C, B, 1, db, 0, stream);
}

int fs_compress_softmax_fwd(const float* s, const float* w, const float* bias, float* xs, int B, int HW, int C,
                            hipStream_t stream) {
  FS_REQUIRE(s && w && bias && xs && B > 0 && HW > 0 && HW <= 16384 && C > 0 && C <= 32);
  hipLaunchKernelGGL(compress_softmax_fwd_kernel, dim3(B), dim3(1024), HW * sizeof(float), stream, s, w, bias, xs, HW, C);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: floats of scratch fs_compress_softmax_bwd needs (one record of C + 1 partial sums per workgroup)
long fs_compress_softmax_bwd_scratch_floats(int B, int C) { return (B > 0 && C > 0) ? (long)B * CS_SLICES * (C + 1) : 0; }

int fs_compress_softmax_bwd(const float* g, const float* xs, const float* s, const float* w, float* ds, float* dw, float* db,
                            int B, int HW, int C, float* scratch, hipStream_t stream) {
  FS_REQUIRE(g && xs && s && w && ds && dw && db && scratch && B > 0 && HW > 0 && C > 0 && C <= 32 && C % 4 == 0 && CS_THREADS % (C / 4) == 0);
  FS_REQUIRE((((size_t)s | (size_t)ds | (size_t)w) & 15) == 0);
  const int nwg = B * CS_SLICES;
  hipLaunchKernelGGL(compress_softmax_bwd_kernel, dim3(nwg), dim3(CS_THREADS), 0, stream, g, xs, s, w, ds, scratch, HW, C);
  FS_LAUNCH_CHECK();
  const int r = fs_slab_reduce(scratch, nwg, C, dw, 0, stream);
  if (r != FS_OK) return r;
  return fs_slab_reduce(scratch + (long)nwg * C, nwg, 1, db, 0, stream);
}

int fs_area_pool_fwd(const float* y, float* out, int B, int H, int W, int hs, int ws, hipStream_t stream) {
  FS_REQUIRE(y && out && B > 0 && H >= hs && W >= ws && hs > 0 && ws > 0 && W <= 16384);
  hipLaunchKernelGGL(area_pool_kernel, dim3(B * hs), dim3(256), W * sizeof(float), stream, y, out, H, W, hs, ws);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: floats behind `stats` (6 statistics + 2 pad + per-workgroup partial records of both passes)
long fs_edge_loss_stats_floats(long n) {
  if (n <= 0) return 0;
  const int G = el_wgs(n);
  return 8 + 4L * G + 2L * (4L * G + 2L * G);
}

int fs_edge_loss_fwd(const float* xs, const float* t, long n, float coef, float* loss, float* stats, hipStream_t stream) {
  FS_REQUIRE(xs && t && loss && stats && n > 0 && (((size_t)stats) & 31) == 0);
  const int G = el_wgs(n);
  hipLaunchKernelGGL(edge_minmax_kernel, dim3(G), dim3(EL_THREADS), 0, stream, xs, t, n, stats);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(edge_sums_kernel, dim3(G), dim3(EL_THREADS), 0, stream, xs, t, n, stats);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(edge_finish_kernel, dim3(1), dim3(64), 0, stream, n, coef, G, loss, stats);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// stats: the buffer fs_edge_loss_fwd filled (its scratch part is written here: not const)
int fs_edge_loss_bwd(const float* xs, const float* t, long n, float coef, const float* gout, float* stats, float* dxs,
                     hipStream_t stream) {
  FS_REQUIRE(xs && t && gout && stats && dxs && n > 0 && (((size_t)stats) & 31) == 0);
  const int G = el_wgs(n);
  hipLaunchKernelGGL(edge_bwd_sums_kernel, dim3(G), dim3(EL_THREADS), 0, stream, xs, t, n, coef, gout, stats);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(edge_bwd_apply_kernel, dim3(G), dim3(EL_THREADS), 0, stream, xs, t, n, coef, gout, stats, dxs);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

static int gg_set_lds(const void* fn, int bytes, unsigned long long& done) {
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
  if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {        // the dynamic-LDS opt-in is a per-device function attribute
    const hipError_t e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e != hipSuccess) return (int)e;
    if (dev >= 0 && dev < 64) done |= 1ull << dev;
  }
  return FS_OK;
}

extern "C++" {
template <int MODE>
static int gauss_grid_fwd_launch(const float* xs, const double* g1d, float* grid, int B, int hs, int ws, int pad, hipStream_t stream) {
  static unsigned long long done = 0ull;
  const int r = gg_set_lds((const void*)gauss_grid_fwd_kernel<MODE>, 3 * GMAX * (int)sizeof(float), done);
  if (r != FS_OK) return r;
  hipLaunchKernelGGL(gauss_grid_fwd_kernel<MODE>, dim3(B * GG_BANDS), dim3(1024), 3 * GMAX * sizeof(float), stream, xs, g1d, grid, hs, ws, pad);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
}  // extern "C++"

// include/fovealseg.h: pad_mode 0 replication / 1 reflect / 2 zero (TRAIN.def_saliency_pad_mode, models/models.py:819-825)
int fs_gauss_grid_fwd_mode(const float* xs, const double* g1d, float* grid, int B, int hs, int ws, int pad, int pad_mode, hipStream_t stream) {
  FS_REQUIRE(xs && g1d && grid && B > 0 && hs > 1 && ws > 1 && hs * ws <= GMAX && pad >= 0 && 2 * pad + 1 <= 256);
  FS_REQUIRE(pad_mode >= PAD_REPLICATION && pad_mode <= PAD_ZERO);
  FS_REQUIRE(pad_mode != PAD_REFLECT || (pad <= hs - 1 && pad <= ws - 1));          // F.pad(mode='reflect') refuses a pad >= the side
  if (pad_mode == PAD_REFLECT) return gauss_grid_fwd_launch<PAD_REFLECT>(xs, g1d, grid, B, hs, ws, pad, stream);
  if (pad_mode == PAD_ZERO) return gauss_grid_fwd_launch<PAD_ZERO>(xs, g1d, grid, B, hs, ws, pad, stream);
  return gauss_grid_fwd_launch<PAD_REPLICATION>(xs, g1d, grid, B, hs, ws, pad, stream);
}

int fs_gauss_grid_fwd(const float* xs, const double* g1d, float* grid, int B, int hs, int ws, int pad, hipStream_t stream) {
  return fs_gauss_grid_fwd_mode(xs, g1d, grid, B, hs, ws, pad, PAD_REPLICATION, stream);
}

// include/fovealseg.h: floats of scratch fs_gauss_grid_bwd needs ((dp, dax, day) per grid point, handed from its first launch to its second)
long fs_gauss_grid_bwd_scratch_floats(int B, int hs, int ws) { return (B > 0 && hs > 0 && ws > 0) ? 3L * B * hs * ws : 0; }

extern "C++" {
template <int MODE>
static int gauss_grid_bwd_launch(const float* xs, const double* g1d, const float* dgrid, float* dxs, int B, int hs, int ws, int pad,
                                 float* scratch, hipStream_t stream) {
  static unsigned long long done1 = 0ull, done2 = 0ull;
  int r = gg_set_lds((const void*)gauss_grid_bwd1_kernel<MODE>, 3 * GMAX * (int)sizeof(float), done1);
  if (r != FS_OK) return r;
  r = gg_set_lds((const void*)gauss_grid_bwd2_kernel<MODE>, 5 * GMAX * (int)sizeof(float), done2);
  if (r != FS_OK) return r;
  hipLaunchKernelGGL(gauss_grid_bwd1_kernel<MODE>, dim3(B * GG_BANDS), dim3(1024), 3 * GMAX * sizeof(float), stream, xs, g1d, dgrid, scratch, hs, ws, pad);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(gauss_grid_bwd2_kernel<MODE>, dim3(B * GG_BANDS), dim3(1024), 5 * GMAX * sizeof(float), stream, scratch, g1d, dxs, hs, ws, pad);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
}  // extern "C++"

int fs_gauss_grid_bwd_mode(const float* xs, const double* g1d, const float* dgrid, float* dxs, int B, int hs, int ws, int pad, int pad_mode,
                           float* scratch, hipStream_t stream) {
  FS_REQUIRE(xs && g1d && dgrid && dxs && scratch && B > 0 && hs > 1 && ws > 1 && hs * ws <= GMAX && pad >= 0 && 2 * pad + 1 <= 256);
  FS_REQUIRE(pad_mode >= PAD_REPLICATION && pad_mode <= PAD_ZERO);
  FS_REQUIRE(pad_mode != PAD_REFLECT || (pad <= hs - 1 && pad <= ws - 1));
  if (pad_mode == PAD_REFLECT) return gauss_grid_bwd_launch<PAD_REFLECT>(xs, g1d, dgrid, dxs, B, hs, ws, pad, scratch, stream);
  if (pad_mode == PAD_ZERO) return gauss_grid_bwd_launch<PAD_ZERO>(xs, g1d, dgrid, dxs, B, hs, ws, pad, scratch, stream);
  return gauss_grid_bwd_launch<PAD_REPLICATION>(xs, g1d, dgrid, dxs, B, hs, ws, pad, scratch, stream);
}

int fs_gauss_grid_bwd(const float* xs, const double* g1d, const float* dgrid, float* dxs, int B, int hs, int ws, int pad,
                      float* scratch, hipStream_t stream) {
  return fs_gauss_grid_bwd_mode(xs, g1d, dgrid, dxs, B, hs, ws, pad, PAD_REPLICATION, scratch, stream);
}

// include/fovealseg.h: fs_grid_upsample_fwd / _bwd
int fs_grid_upsample_fwd(const float* grid, float* out, int B, int h, int w, int H, int W, hipStream_t stream) {
  FS_REQUIRE(grid && out && B > 0 && h > 0 && w > 0 && H >= h && W >= w);
  hipLaunchKernelGGL(grid_upsample_fwd_kernel, dim3(cdiv((long)B * H * W, 256)), dim3(256), 0, stream, reinterpret_cast<const float2*>(grid),
                     reinterpret_cast<float2*>(out), B, h, w, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
int fs_grid_upsample_bwd(const float* g, float* dgrid, int B, int h, int w, int H, int W, hipStream_t stream) {
  FS_REQUIRE(g && dgrid && B > 0 && h > 0 && w > 0 && H >= h && W >= w && H % h == 0 && W % w == 0);      // integer factors
  hipLaunchKernelGGL(grid_upsample_bwd_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, reinterpret_cast<const float2*>(g),
                     reinterpret_cast<float2*>(dgrid), B, h, w, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_grid_sample_fwd(const float* x, const float* grid, float* out, int B, int C, int H, int W, int h, int w, int nhwc_out,
                       hipStream_t stream) {
  FS_REQUIRE(x && grid && out && B > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(grid_sample_fwd_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, x, grid, out, B, C, H, W, h, w, nhwc_out);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_grid_sample_label(const float* y, const float* grid, long long* label, float* ysamp, int B, int H, int W, int h, int w,
                         hipStream_t stream) {
  FS_REQUIRE(y && grid && label && B > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(grid_sample_label_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, y, grid, label, ysamp, B, H, W, h, w);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_grid_sample_bwd_grid(const float* gout, const float* x, const float* grid, float* dgrid, int B, int C, int H, int W, int h,
                            int w, int nhwc, hipStream_t stream) {
  FS_REQUIRE(gout && x && grid && dgrid && B > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipLaunchKernelGGL(grid_sample_bwd_grid_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, gout, x, grid, dgrid, B, C, H, W, h, w, nhwc);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_grid_sample_bwd_input(const float* gout, const float* grid, float* dx, int B, int C, int H, int W, int h, int w, int nhwc,
                             hipStream_t stream) {
  FS_REQUIRE(gout && grid && dx && B > 0 && C > 0 && H > 0 && W > 0 && h > 0 && w > 0);
  hipError_t e = hipMemsetAsync(dx, 0, sizeof(float) * (size_t)B * C * H * W, stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(grid_sample_bwd_input_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, gout, grid, dx, B, C, H, W, h, w, nhwc);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_inverse_grid(const float* grid, int* owner, float* grid_inv, int B, int h, int w, int Hs, int Ws, hipStream_t stream) {
  FS_REQUIRE(grid && owner && grid_inv && B > 0 && h > 0 && w > 0 && Hs > 0 && Ws > 0 && (long)h * w < 2147483647L);
  const long n = (long)B * Hs * Ws;
  hipError_t e = hipMemsetAsync(owner, 0xFF, sizeof(int) * (size_t)n, stream);      // -1
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(inverse_owner_kernel, dim3(cdiv((long)B * h * w, 256)), dim3(256), 0, stream, grid, owner, B, h * w, Hs, Ws);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(inverse_grid_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, owner, grid_inv, n, h, w);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_fill_nearest(float* vals, const int* owner, int* scratch, int B, int C, int Hs, int Ws, hipStream_t stream) {
  FS_REQUIRE(vals && owner && scratch && B > 0 && C > 0 && Hs > 0 && Ws > 0 && (long)Hs * Ws < 2147483647L);
  const long per = (long)Hs * Ws, n = (long)B * per;
  int* rowx = scratch;          // [B*Hs*Ws]
  int* src = scratch + n;       // [B*Hs*Ws]
  FS_REQUIRE(Ws <= 16384);     // one row of column indices in LDS
  hipLaunchKernelGGL(fill_row_nearest_kernel, dim3((unsigned)((long)B * Hs)), dim3(256), (size_t)Ws * sizeof(int), stream, owner, rowx, Ws);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(fill_col_nearest_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, rowx, src, n, Hs, Ws);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(fill_copy_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, vals, owner, src, C, per, n);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_inverse_index_maps(const float* grid, long long* u, long long* v, long n, int H, int W, hipStream_t stream) {
  FS_REQUIRE(grid && u && v && n > 0 && H > 0 && W > 0);
  hipLaunchKernelGGL(inverse_index_kernel, dim3(cdiv(n, 256)), dim3(256), 0, stream, grid, u, v, n, H, W);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // extern "C"
