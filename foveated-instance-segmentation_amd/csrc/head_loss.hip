// C1 head tail + per-pixel losses:
//   mask branch  m = sigmoid(conv1x1(x, 240->1) + b) - 0.5            (models/model_utils.py:293-298)
//   pred[:, :K-1] = c broadcast, pred[:, K-1] = c[K-1] * m            (models/model_utils.py:300-306)
//   FocalLoss(gamma) + DiceLoss('multiclass') + 4 IoU-style accuracies (models/models.py:87-120,
//   378-474,1057-1078; Dice = pytorch_toolbelt 0.8.0 restated, see oracle header)
#include "common.h"

namespace {

// one wave per pixel: dot over C (NHWC row), sigmoid - 0.5
__global__ __launch_bounds__(256) void mask_head_fwd_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, float* __restrict__ m, long npix,
                                                            int C) {
  const int lane = threadIdx.x & 63;
  const long wave = ((long)blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  const long nwaves = ((long)gridDim.x * blockDim.x) >> 6;
  for (long p = wave; p < npix; p += nwaves) {
    float acc = 0.f;
    for (int c = 4 * lane; c < C; c += 256) {
      const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + c);
      const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
      acc += v.x * ww.x + v.y * ww.y + v.z * ww.z + v.w * ww.w;
    }
    acc = wave_sum(acc);
    if (lane == 0) m[p] = 1.f / (1.f + expf(-(acc + bias[0]))) - 0.5f;
  }
}

// dlogit = dm * s(1-s), s = m + 0.5;  dx[p][c] = dlogit*w[c];  dw[c] += sum_p dlogit*x[p][c];  db += sum dlogit
__global__ __launch_bounds__(256) void mask_head_bwd_kernel(const float* __restrict__ dm, const float* __restrict__ m,
                                                            const float* __restrict__ x, const float* __restrict__ w,
                                                            float* __restrict__ dx, float* __restrict__ part /* [blocks][C], then [blocks] */,
                                                            long npix, int C, int pix_per_block) {
  extern __shared__ float dws[];   // [4 waves][C] floats: the waves' partial sums, added in wave order (no LDS atomics)
  __shared__ float dbs[4];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const long p0 = (long)blockIdx.x * pix_per_block;
  long p1 = p0 + pix_per_block; if (p1 > npix) p1 = npix;
  f32x4 dwl[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) dwl[j] = f32x4{0, 0, 0, 0};
  float dbl = 0.f;
  for (long p = p0 + wv; p < p1; p += 4) {
    const float s = m[p] + 0.5f;
    const float dl = dm[p] * s * (1.f - s);
    dbl += dl;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int c = 4 * lane + 256 * j;
      if (c < C) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + p * C + c);
        const f32x4 ww = *reinterpret_cast<const f32x4*>(w + c);
        *reinterpret_cast<f32x4*>(dx + p * C + c) = dl * ww;
        dwl[j] += dl * v;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 4 * lane + 256 * j;
    if (c < C) *reinterpret_cast<f32x4*>(&dws[wv * C + c]) = dwl[j];
  }
  // dl is wave-uniform (one pixel per wave and round): lane 0 holds the wave's sum already
  if (lane == 0) dbs[wv] = dbl;
  __syncthreads();
  // this workgroup's record, summed in block order by fs_slab_reduce
  for (int c = threadIdx.x; c < C; c += blockDim.x) part[(long)blockIdx.x * C + c] = ((dws[c] + dws[C + c]) + dws[2 * C + c]) + dws[3 * C + c];
  if (threadIdx.x == 0) part[(long)gridDim.x * C + blockIdx.x] = ((dbs[0] + dbs[1]) + dbs[2]) + dbs[3];
}

// pred (B,K,HW) NCHW from class logits c (B,K) and mask m (B,HW)
__global__ __launch_bounds__(256) void pred_assemble_fwd_kernel(const float* __restrict__ cls, const float* __restrict__ m,
                                                                float* __restrict__ pred, int B, int K, int HW) {
  const long total = (long)B * K * HW;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int p = (int)(i % HW);
    const int k = (int)((i / HW) % K);
    const int b = (int)(i / ((long)HW * K));
    const float c = cls[b * K + k];
    pred[i] = (k == K - 1) ? c * m[(long)b * HW + p] : c;
  }
}

// one block per (b,k): dcls[b,k] = sum_p dpred (* m for k = K-1); block k=K-1 also writes dm
__global__ __launch_bounds__(256) void pred_assemble_bwd_kernel(const float* __restrict__ dpred, const float* __restrict__ cls,
                                                                const float* __restrict__ m, float* __restrict__ dcls,
                                                                float* __restrict__ dm, int K, int HW) {
  __shared__ float red[16];
  const int b = blockIdx.x / K, k = blockIdx.x - b * K;
  const float* d = dpred + ((long)b * K + k) * HW;
  float s = 0.f;
  if (k == K - 1) {
    const float c = cls[b * K + k];
    for (int p = threadIdx.x; p < HW; p += blockDim.x) {
      const float g = d[p];
      s += g * m[(long)b * HW + p];
      dm[(long)b * HW + p] = g * c;
    }
  } else {
    for (int p = threadIdx.x; p < HW; p += blockDim.x) s += d[p];
  }
  s = block_sum<float>(s, red);
  if (threadIdx.x == 0) dcls[b * K + k] = s;
}

// ---- segmentation loss ---------------------------------------------------------------------
// accumulators (double): [0..K) sum_p p_k | [K..2K) sum_p p_k*[gt==k] | [2K..3K) count[gt==k] |
//   [3K] focal sum | then per image 6 counters: cls_fg, bin_fg, union_fg, cls_bg, bin_bg, union_bg
constexpr int KMAX = 64;

__global__ __launch_bounds__(256) void seg_loss_fwd_kernel(const float* __restrict__ pred, const long long* __restrict__ gt,
                                                           double* __restrict__ accum, int K, int HW, int blocks_per_img,
                                                           float gamma) {
  // sp: per-wave class sums, added in wave order; si: sum of the true class's probability in 2^-40 fixed point (integer adds commute:
  // no dependence on which lane gets to the LDS first); sc: counts (exact in fp32)
  __shared__ float sp[4][KMAX], sc[KMAX];
  __shared__ unsigned long long si[KMAX];
  __shared__ double dred[16];
  const int b = blockIdx.x / blocks_per_img, chunk = blockIdx.x - b * blocks_per_img;
  for (int k = threadIdx.x; k < KMAX; k += blockDim.x) { si[k] = 0ull; sc[k] = 0.f; }
  __syncthreads();
  const float* pb = pred + (long)b * K * HW;
  const int bg = K - 1;
  double focal = 0.0;
  float cnt[6] = {0, 0, 0, 0, 0, 0};
  // per-class probability sums stay in registers over the thread's pixels (KMAX floats per lane) and are reduced across the
  // wave ONCE after the loop: one LDS atomic per wave and class, instead of a 6-step shuffle tree per class and pixel
  const int lane = threadIdx.x & 63;
  float spl[KMAX];
#pragma unroll
  for (int k = 0; k < KMAX; ++k) spl[k] = 0.f;
  for (int p0 = chunk * blockDim.x; p0 < HW; p0 += blocks_per_img * blockDim.x) {
    const int p = p0 + threadIdx.x;
    if (p >= HW) continue;
    const int t = (int)gt[(long)b * HW + p];
    // the pixel's K logits are loaded ONCE, all loads in flight together (a runtime-K loop of load -> compare made every one of
    // the 3 x 51 loads wait for the previous: 36 us of pure latency per pixel), and the three passes run on registers
    float v[KMAX];
#pragma unroll
    for (int k = 0; k < KMAX; ++k) v[k] = k < K ? pb[(long)k * HW + p] : -INFINITY;
    float mx = -INFINITY; int am = 0;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (v[k] > mx) { mx = v[k]; am = k; }
    float se = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) if (k < K) se += expf(v[k] - mx);
    const float lse = logf(se);
    float vt = 0.f;
#pragma unroll
    for (int k = 0; k < KMAX; ++k) {
      if (k < K) {
        const float pk = expf(v[k] - mx - lse);
        spl[k] += pk;
        if (k == t) { atomicAdd(&si[k], (unsigned long long)((double)pk * 1099511627776.0)); atomicAdd(&sc[k], 1.f); vt = v[k]; }
      }
    }
    const float logpt = vt - mx - lse;
    const float pt = expf(logpt);
    focal += (double)(-powf(1.f - pt, gamma) * logpt);
    const bool vg = t < bg, vp = am < bg, bgg = t == bg, bgp = am == bg, eq = am == t;
    cnt[0] += (vg && eq); cnt[1] += (vg && (vg == vp)); cnt[2] += (vg || vp);
    cnt[3] += (bgg && eq); cnt[4] += (bgg && (bgg == bgp)); cnt[5] += (bgg || bgp);
  }
#pragma unroll
  for (int k = 0; k < KMAX; ++k) {
    if (k < K) {
      const float ws = wave_sum(spl[k]);
      if (lane == 0) sp[threadIdx.x >> 6][k] = ws;
    }
  }
  __syncthreads();
  // this workgroup's record [3K + 1 + 6]: plain stores, summed by the finalize kernel -- no zero-initialisation of the scratch and no
  // same-address double atomics from every workgroup
  double* rec = accum + (long)blockIdx.x * (3 * K + 7);
  for (int k = threadIdx.x; k < K; k += blockDim.x) {
    rec[k] = (double)(((sp[0][k] + sp[1][k]) + sp[2][k]) + sp[3][k]);
    rec[K + k] = (double)si[k] * (1.0 / 1099511627776.0);
    rec[2 * K + k] = (double)sc[k];
  }
  focal = block_sum<double>(focal, dred);
  if (threadIdx.x == 0) rec[3 * K] = focal;
  for (int j = 0; j < 6; ++j) {
    const double v = block_sum<double>((double)cnt[j], dred);
    if (threadIdx.x == 0) rec[3 * K + 1 + j] = v;
  }
}

// out[0]=dice+focal, out[1]=focal, out[2]=dice, out[3..6]=acc, acc_bin_fg, acc_cls_fbg, acc_bin_fbg
// coef (2K floats): A_k = present/K * 2 I/Kc^2, B_k = -present/K * 2/Kc  (for the backward)
// accum = [B * blocks_per_img] records of (3K + 7) doubles (seg_loss_fwd_kernel); one workgroup sums them and finalises
__global__ __launch_bounds__(1024) void seg_loss_finalize_kernel(const double* __restrict__ accum, int B, int K, int HW, int blocks_per_img,
                                                                 float eps, float* __restrict__ out, float* __restrict__ coef) {
  __shared__ double tot[3 * KMAX + 1];
  __shared__ double img[4][16];
  const int R = 3 * K + 7, nrec = B * blocks_per_img;
  // column sums over the records: wave w takes entries w, w + 16, ...; its lanes stride over the records
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
  for (int e = wv; e < 3 * K + 1; e += nw) {
    double s = 0.0;
    for (int i = lane; i < nrec; i += 64) s += accum[(long)i * R + e];
    s = wave_sum_d(s);
    if (lane == 0) tot[e] = s;
  }
  // per-image accuracies: thread b sums the six counters of its image's records
  double a[4] = {0, 0, 0, 0};
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    double c[6] = {0, 0, 0, 0, 0, 0};
    for (int j = 0; j < blocks_per_img; ++j)
      for (int q = 0; q < 6; ++q) c[q] += accum[(long)(b * blocks_per_img + j) * R + 3 * K + 1 + q];
    const float ufg = (float)c[2] + 1e-10f, ubg = (float)c[5] + 1e-10f;
    const float cls_fg = (float)c[0] / ufg, bin_fg = (float)c[1] / ufg, cls_bg = (float)c[3] / ubg, bin_bg = (float)c[4] / ubg;
    a[0] += cls_fg; a[1] += bin_fg; a[2] += cls_fg * 0.5f + cls_bg * 0.5f; a[3] += bin_fg * 0.5f + bin_bg * 0.5f;
  }
  for (int j = 0; j < 4; ++j) a[j] = block_sum<double>(a[j], img[j]);
  __syncthreads();
  double lk = 0.0;
  if (threadIdx.x < K) {
    const int k = threadIdx.x;
    const float I = (float)tot[K + k];
    const float card = (float)tot[k] + (float)tot[2 * K + k];
    const bool present = tot[2 * K + k] > 0.0;
    const float den = fmaxf(card, eps);
    lk = present ? (double)(1.f - 2.f * I / den) : 0.0;
    float A = 0.f, Bc = 0.f;
    if (present) {
      Bc = -2.f / den / (float)K;
      A = (card > eps) ? 2.f * I / (den * den) / (float)K : 0.f;
    }
    coef[k] = A; coef[K + k] = Bc;
  }
  const double dice = block_sum<double>(lk, img[0]);
  if (threadIdx.x != 0) return;
  const float dl = (float)(dice / (double)K);
  const float fl = (float)(tot[3 * K] / ((double)B * HW));
  out[0] = dl + fl; out[1] = fl; out[2] = dl;
  for (int j = 0; j < 4; ++j) out[3 + j] = (float)(a[j] / (double)B);
}

// dpred = gout * (d focal + d dice)
__global__ __launch_bounds__(256) void seg_loss_bwd_kernel(const float* __restrict__ pred, const long long* __restrict__ gt,
                                                           const float* __restrict__ coef, const float* __restrict__ gout,
                                                           float* __restrict__ dpred, int B, int K, int HW, float gamma) {
  __shared__ float cA[KMAX], cB[KMAX];
  for (int k = threadIdx.x; k < K; k += blockDim.x) { cA[k] = coef[k]; cB[k] = coef[K + k]; }
  __syncthreads();
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (long)B * HW) return;
  const int b = (int)(i / HW), p = (int)(i - (long)b * HW);
  const float* pb = pred + (long)b * K * HW;
  float* db = dpred + (long)b * K * HW;
  const int t = (int)gt[i];
  const float go = gout[0];
  float mx = -INFINITY;
  for (int k = 0; k < K; ++k) mx = fmaxf(mx, pb[(long)k * HW + p]);
  float se = 0.f;
  for (int k = 0; k < K; ++k) se += expf(pb[(long)k * HW + p] - mx);
  const float lse = logf(se);
  float dotpq = 0.f;
  for (int k = 0; k < K; ++k) {
    const float pk = expf(pb[(long)k * HW + p] - mx - lse);
    dotpq += pk * (cA[k] + (k == t ? cB[k] : 0.f));
  }
  const float pt = expf(pb[(long)t * HW + p] - mx - lse);
  const float fw = powf(1.f - pt, gamma) / (float)((long)B * HW);   // focal: -(fw) * ([k==t] - p_k)
  for (int k = 0; k < K; ++k) {
    const float pk = expf(pb[(long)k * HW + p] - mx - lse);
    const float q = cA[k] + (k == t ? cB[k] : 0.f);
    const float dd = pk * (q - dotpq);
    const float df = -fw * ((k == t ? 1.f : 0.f) - pk);
    db[(long)k * HW + p] = go * (dd + df);
  }
}

}  // namespace

extern "C" {

int fs_mask_head_fwd(const float* x, const float* w, const float* bias, float* m, long npix, int C, hipStream_t stream) {
  FS_REQUIRE(x && w && bias && m && npix > 0 && C > 0 && C % 4 == 0 && C <= 1024);
  int blocks = cdiv(npix, 4); if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(mask_head_fwd_kernel, dim3(blocks), dim3(256), 0, stream, x, w, bias, m, npix, C);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

constexpr int MASK_HEAD_PPB = 256;      // pixels per workgroup of mask_head_bwd_kernel
long fs_mask_head_bwd_scratch_floats(long npix, int C) { return (npix > 0 && C > 0) ? (long)cdiv(npix, MASK_HEAD_PPB) * (C + 1) : 0; }

// scratch: fs_mask_head_bwd_scratch_floats(npix, C) floats (per-workgroup partial sums of dw and db, added in workgroup order)
int fs_mask_head_bwd(const float* dm, const float* m, const float* x, const float* w, float* dx, float* dw, float* db, long npix,
                     int C, float* scratch, hipStream_t stream) {
  FS_REQUIRE(dm && m && x && w && dx && dw && db && scratch && npix > 0 && C > 0 && C % 4 == 0 && C <= 1024);
  const int nblk = cdiv(npix, MASK_HEAD_PPB);
  hipLaunchKernelGGL(mask_head_bwd_kernel, dim3(nblk), dim3(256), 4 * C * sizeof(float), stream, dm, m, x, w, dx, scratch,
                     npix, C, MASK_HEAD_PPB);
  FS_LAUNCH_CHECK();
  const int r = fs_slab_reduce(scratch, nblk, C, dw, 0, stream);
  if (r != FS_OK) return r;
  return fs_slab_reduce(scratch + (long)nblk * C, nblk, 1, db, 0, stream);
}

int fs_pred_assemble_fwd(const float* cls, const float* m, float* pred, int B, int K, int HW, hipStream_t stream) {
  FS_REQUIRE(cls && m && pred && B > 0 && K > 1 && HW > 0);
  int blocks = cdiv((long)B * K * HW, 256); if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(pred_assemble_fwd_kernel, dim3(blocks), dim3(256), 0, stream, cls, m, pred, B, K, HW);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_pred_assemble_bwd(const float* dpred, const float* cls, const float* m, float* dcls, float* dm, int B, int K, int HW,
                         hipStream_t stream) {
  FS_REQUIRE(dpred && cls && m && dcls && dm && B > 0 && K > 1 && HW > 0);
  hipLaunchKernelGGL(pred_assemble_bwd_kernel, dim3(B * K), dim3(256), 0, stream, dpred, cls, m, dcls, dm, K, HW);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// accum: (3K + 1 + 6B) doubles of scratch; out: 7 floats; coef: 2K floats kept for the backward
int fs_seg_loss_fwd(const float* pred, const long long* gt, int B, int K, int HW, float gamma, float eps, double* accum,
                    float* out, float* coef, hipStream_t stream) {
  FS_REQUIRE(pred && gt && accum && out && coef && B > 0 && K > 1 && K <= KMAX && HW > 0);
  const int bpi = cdiv(HW, 1024);           // = the record count the header documents: B * ceil(HW / 1024)
  hipLaunchKernelGGL(seg_loss_fwd_kernel, dim3(B * bpi), dim3(256), 0, stream, pred, gt, accum, K, HW, bpi, gamma);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(seg_loss_finalize_kernel, dim3(1), dim3(1024), 0, stream, accum, B, K, HW, bpi, eps, out, coef);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_seg_loss_bwd(const float* pred, const long long* gt, const float* coef, const float* gout, float* dpred, int B, int K,
                    int HW, float gamma, hipStream_t stream) {
  FS_REQUIRE(pred && gt && coef && gout && dpred && B > 0 && K > 1 && K <= KMAX && HW > 0);
  hipLaunchKernelGGL(seg_loss_bwd_kernel, dim3(cdiv((long)B * HW, 256)), dim3(256), 0, stream, pred, gt, coef, gout, dpred, B, K, HW,
                     gamma);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // extern "C"
