// 2-D convolution entry points of the C ABI (fs_conv2d_fwd / _fwd_stats / _bwd_data / _bwd_weight) and kernel selection.
// NHWC fp32 activations, RSCK weights ([R][S][Cin][Cout]).  Replaces ATen conv2d / convolution_backward on the path
// (reference call sites: models/hrnetv2_nodownsp.py:26-29,72-78,190-216,278-282,327-346;
//  saliency_network.py:308-310; models/model_utils.py:6-13,228-232,260).
//
//   forward   : Y[m][n]  = sum_{tap,ci} X[pix(m)+tap][ci] * W[tap][ci][n]          M=B*Ho*Wo N=Cout
//   bwd-data  : dX[m][n] = sum_{tap,co} dY[(pix(m)+pad-tap)/s][co] * W[tap][n][co] M=B*H*W   N=Cin
//   bwd-weight: dW[tap][ci][co] = sum_pix X[pix+tap][ci] * dY[pix][co]             K=B*Ho*Wo (split)
//
// Which kernel runs (precision mode g_conv_precision: 0 = f32, 1 = bf16x3, 2 = f16x2):
//   3x3 stride 1 pad 1, aligned channels, modes 1/2, scratch given : conv_halo.hip (fwd, bwd-data), conv_wgrad.hip (bwd-weight)
//   other multi-tap filters / strided bwd-data sub-problems, modes 1/2 : conv_tapset.hip; bwd-weight per tap class: conv_wgrad.hip
//   1x1, stride >= filter size, single-tap sub-problems, modes 1/2 : conv_igemm_split_kernel<P> (this file)
//   everything in mode 0                                           : conv_igemm_affine_kernel, conv_wgrad_taps_kernel (fp32 MFMA)
//   channel counts that are not multiples of 4                     : conv_igemm_kernel, conv_wgrad_kernel (generic)
// The kernels in this file share one tiling: 256 threads = 4 waves, workgroup tile 128 (pixels) x 64 (channels), K-step 32
// inside one filter tap; each wave owns 32 x 64 = two 32x32 accumulators; operands are staged global -> registers -> LDS
// with the next stage's loads in flight under the MFMAs.
#include "common.h"
#include "conv_kernels.h"
#include "conv_split.h"
#include <stdlib.h>
#include <string.h>

namespace {

constexpr int BM = 128, BN = 64, BK = 32;
constexpr int A_LD = BK + 1;   // odd leading dimension: conflict-free ds_read_b32 of A[row][k]

// Raw buffer descriptor: 32-bit byte offsets, hardware range check (out-of-range lanes read 0), so
// padding taps / channel tails need no branch -- the lane just gets an offset past num_records.
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
constexpr unsigned OOB = 0xFFFFFFF0u;
__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ f32x4 buf_load4(__amdgpu_buffer_rsrc_t r, unsigned byte_off) {
  u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(r, (int)byte_off, 0, 0);
  return __builtin_bit_cast(f32x4, v);
}

struct ConvArgs {
  const float* src;   // fwd: X (B,Hs,Ws,Cs)      bwd-data: dY
  const float* w;     // [R][S][Cin][Cout]
  const float* bias;  // fwd only, may be null
  float* dst;         // fwd: Y (B,Hd,Wd,Cd)      bwd-data: dX
  int B, Hs, Ws, Cs, Hd, Wd, Cd;
  int R, S, stride, pad, dil;
  int transposed;     // 0 = forward, 1 = bwd-data
  float drop_scale;   // 1/(1-p)
  uint32_t drop_thresh, drop_key;   // thresh 0 = no dropout
  hipStream_t stream_ = nullptr;    // host only
  float* stats_ = nullptr;          // host only: BN partial-sum slab (affine forward)
  const FsBnSums* bn_ = nullptr;    // host only: bwd-data writes the consumer BatchNorm's backward sums into stats_ (F(2,3) kernels)
  void* ws_ = nullptr;              // host only: caller's scratch for the pre-split weight pack (may be null)
  long ws_bytes_ = 0;
  const unsigned* w_amax_ = nullptr;   // host only: max|w| bits kept by the caller (f16x2 mode), may be null
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv_igemm_kernel(ConvArgs a) {
  __shared__ float As[BM * A_LD];
  __shared__ float Bs[BK * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long M = (long)a.B * a.Hd * a.Wd;
  const long m0 = (long)blockIdx.x * BM;
  const int n0 = blockIdx.y * BN;

  // ---- per-thread A rows: row = (tid>>3) + 32*i, channel quad q = tid&7 -------------------
  const int q = tid & 7, arow = tid >> 3;
  int pb[4], py[4], px[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    long m = m0 + arow + 32 * i;
    if (m < M) {
      int b = (int)(m / ((long)a.Hd * a.Wd));
      int rem = (int)(m - (long)b * a.Hd * a.Wd);
      pb[i] = b; py[i] = rem / a.Wd; px[i] = rem - py[i] * a.Wd;
    } else {
      pb[i] = -1; py[i] = 0; px[i] = 0;
    }
  }
  // ---- per-thread B slots ---------------------------------------------------------------
  //   forward : tile [k][n], n contiguous in memory: n4 = tid&15, k = (tid>>4)+16*i
  //   bwd-data: tile [n][k], k contiguous in memory: k4 = tid&7,  n = (tid>>3)+32*i
  f32x4 ra[4], rb[2];
  const int nchunk = (a.Cs + BK - 1) / BK;
  const int nstage = a.R * a.S * nchunk;

  auto load_stage = [&](int st) {
    const int tap = st / nchunk, c0 = (st - tap * nchunk) * BK;
    const int tr = tap / a.S, ts = tap - tr * a.S;
    const int c = c0 + 4 * q;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (pb[i] >= 0) {
        int iy, ix;
        bool ok;
        if (!a.transposed) {
          iy = py[i] * a.stride - a.pad + tr * a.dil;
          ix = px[i] * a.stride - a.pad + ts * a.dil;
          ok = (iy >= 0) & (iy < a.Hs) & (ix >= 0) & (ix < a.Ws);
        } else {
          int ty = py[i] + a.pad - tr * a.dil, tx = px[i] + a.pad - ts * a.dil;
          ok = (ty >= 0) & (tx >= 0);
          if (a.stride > 1) {
            ok = ok & (ty % a.stride == 0) & (tx % a.stride == 0);
            iy = ty / a.stride; ix = tx / a.stride;
          } else { iy = ty; ix = tx; }
          ok = ok & (iy < a.Hs) & (ix < a.Ws);
        }
        if (ok) {
          const float* p = a.src + (((long)pb[i] * a.Hs + iy) * a.Ws + ix) * a.Cs + c;
          if (VEC) {
            if (c < a.Cs) v = *reinterpret_cast<const f32x4*>(p);
          } else {
            if (c + 0 < a.Cs) v.x = p[0];
            if (c + 1 < a.Cs) v.y = p[1];
            if (c + 2 < a.Cs) v.z = p[2];
            if (c + 3 < a.Cs) v.w = p[3];
          }
        }
      }
      ra[i] = v;
    }
    if (!a.transposed) {
      const int n = n0 + 4 * (tid & 15);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int k = c0 + (tid >> 4) + 16 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (k < a.Cs) {
          const float* p = a.w + ((long)tap * a.Cs + k) * a.Cd + n;
          if (VEC) {
            if (n < a.Cd) v = *reinterpret_cast<const f32x4*>(p);
          } else {
            if (n + 0 < a.Cd) v.x = p[0];
            if (n + 1 < a.Cd) v.y = p[1];
            if (n + 2 < a.Cd) v.z = p[2];
            if (n + 3 < a.Cd) v.w = p[3];
          }
        }
        rb[i] = v;
      }
    } else {
      const int k = c0 + 4 * (tid & 7);
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = n0 + (tid >> 3) + 32 * i;
        f32x4 v = {0.f, 0.f, 0.f, 0.f};
        if (n < a.Cd) {
          const float* p = a.w + ((long)tap * a.Cd + n) * a.Cs + k;   // W[tap][ci=n][co=k]
          if (VEC) {
            if (k < a.Cs) v = *reinterpret_cast<const f32x4*>(p);
          } else {
            if (k + 0 < a.Cs) v.x = p[0];
            if (k + 1 < a.Cs) v.y = p[1];
            if (k + 2 < a.Cs) v.z = p[2];
            if (k + 3 < a.Cs) v.w = p[3];
          }
        }
        rb[i] = v;
      }
    }
  };

  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      float* p = &As[(arow + 32 * i) * A_LD + 4 * q];
      p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w;
    }
    if (!a.transposed) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<f32x4*>(&Bs[((tid >> 4) + 16 * i) * BN + 4 * (tid & 15)]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = (tid >> 3) + 32 * i, k = 4 * (tid & 7);
        Bs[(k + 0) * BN + n] = rb[i].x;
        Bs[(k + 1) * BN + n] = rb[i].y;
        Bs[(k + 2) * BN + n] = rb[i].z;
        Bs[(k + 3) * BN + n] = rb[i].w;
      }
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const int l31 = lane & 31, lh = lane >> 5;
  const float* Ap = &As[(wave * 32 + l31) * A_LD + lh];
  const float* Bp = &Bs[lh * BN + l31];

  load_stage(0);
  for (int st = 0; st < nstage; ++st) {
    __syncthreads();                 // previous stage's LDS reads are done
    store_stage();
    __syncthreads();
    if (st + 1 < nstage) load_stage(st + 1);   // in flight under the MFMAs below
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) {
      const float av = Ap[2 * kk];
      const float b0 = Bp[2 * kk * BN];
      const float b1 = Bp[2 * kk * BN + 32];
      acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b0, acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av, b1, acc1, 0, 0, 0);
    }
  }

  // ---- epilogue: bias, dropout, store ----------------------------------------------------
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + 32 * t + l31;
    if (n >= a.Cd) continue;
    const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (m >= M) continue;
      float v = (t == 0 ? acc0[r] : acc1[r]) + bv;
      const long e = m * a.Cd + n;
      if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
      a.dst[e] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Fast path ("affine"): forward at any stride and bwd-data at stride 1, channels multiple of 4.
// The pixel of tap (r,s) is a fixed offset from the tap-(0,0) pixel, so every per-stage address is
//   row_offset (per thread, hoisted) + uniform scalar;  validity of the 9 taps is a per-row bit mask.
// LDS is double-buffered: one barrier per K-step; the next tile is written to the other buffer in
// the middle of the MFMA block and the tile after that is requested from global right behind it.
//   MT = 32-row MFMA tiles per wave along M: workgroup tile (128*MT) x 64.
// ------------------------------------------------------------------------------------------
struct AffArgs {
  const float* src; const float* w; const float* bias; float* dst;
  int B, Hs, Ws, Cs, Hd, Wd, Cd;
  int R, S, stride, pad, dil;
  int transposed;
  float drop_scale; uint32_t drop_thresh, drop_key;
  unsigned src_bytes, w_bytes, dst_bytes;   // dst_bytes = 0: Y / dX is 4 GB or more, 64-bit stores
  // output sub-grid (stride>1 bwd-data is run as stride^2 dense sub-problems, one per output parity
  // class, each with its own tap subset): loop grid Hq x Wq, dst pixel = (py*os+oy0, px*os+ox0),
  // taps r = r0 + tstep*tr' (nR of them), s likewise; source row of tap' (0,0) = py + cy.
  int Hq, Wq, os, oy0, ox0, r0, s0, tstep, nR, nS, cy, cx;
  int nx, ny;     // tile grid (1-D launch, XCD-aware remap)
  float* stats;   // optional [nx][Cd][2] per-workgroup column sums / sums of squares of the STORED values
};

static thread_local hipStream_t a_stream = nullptr;   // host: stream of the launch being issued
// 0 = fp32 MFMA (v_mfma_f32_32x32x2_f32), 1 = split-precision bf16x3 (six v_mfma_f32_32x32x16_bf16 per product),
// 2 = f16x2 (three v_mfma_f32_32x32x16_f16 per product on scaled operands) in the 3x3 stride-1 kernels, bf16x3 elsewhere.
// Default 1 = bf16x3: operands carry the reference's full 24 significand bits (the mode bench.py's headline and the parity claims are
// quoted in).  FS_CONV_PRECISION=f32|bf16x3|f16x2 read ONCE at load time; fs_set_conv_precision() overrides.
static int g_conv_precision = [] {
  const char* e = getenv("FS_CONV_PRECISION");
  if (e && strcmp(e, "f32") == 0) return 0;
  if (e && strcmp(e, "f16x2") == 0) return 2;
  return 1;                                            // bf16x3
}();

// Deterministic mode (fs_set_deterministic; FS_DETERMINISTIC=1 read once at load): every reduction whose order depends on workgroup
// arrival -- the split-K atomics of the bwd-weight kernels -- is replaced by per-split partial tiles summed in index order.
static int g_deterministic = [] { const char* e = getenv("FS_DETERMINISTIC"); return (e && e[0] == '1') ? 1 : 0; }();

template <int MT>
__global__ __launch_bounds__(256) void conv_igemm_affine_kernel(AffArgs a) {
  constexpr int TM = 128 * MT;                    // workgroup rows
  constexpr int NR = 4 * MT;                      // A rows per thread
  __shared__ float As[2][TM * A_LD];
  __shared__ float Bs[2][BK * BN];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long M = (long)a.B * a.Hq * a.Wq;
  // XCD-aware tile order: workgroups are dealt round-robin over the 8 XCDs, so give each XCD a
  // contiguous range of tiles (neighbouring pixel tiles share halo rows, the ny channel tiles of one
  // pixel tile share the whole A tile -> they meet in the same 4 MB L2).  Bijective for any count.
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
  const long m0 = (long)(wg / a.ny) * TM;
  const int n0 = (wg % a.ny) * BN;
  const int q = tid & 7, arow = tid >> 3;
  const int sgn = a.transposed ? -1 : 1;
  const int ntap = a.nR * a.nS;
  const bool pointwise = ntap == 1 && a.Hq == a.Hs && a.Wq == a.Ws && a.os == 1 &&
                         (a.transposed ? (a.cy == 0 && a.cx == 0) : (a.stride == 1 && a.pad == 0));

  // ---- hoisted per-row state: element offset of tap (0,0) and the tap validity mask -----------
  int roff[NR];
  uint32_t rmask[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const long m = m0 + arow + 32 * i;
    roff[i] = 0; rmask[i] = 0u;
    if (m < M && pointwise) {
      // single tap landing on the pixel itself (1x1 stride-1 convs, forward and bwd-data): source row = loop row, no (b, y, x)
      // decomposition -- the two integer divisions per row were ~30 % of this kernel's VALU instructions on 64->256 @ 80x80
      roff[i] = (int)m * a.Cs + 4 * q;
      rmask[i] = 1u;
    } else if (m < M) {
      const unsigned mu = (unsigned)m, hw = (unsigned)(a.Hq * a.Wq);       // M < 2^30: the source is below 4 GB (aligned_ok)
      const int b = (int)(mu / hw);
      const int rem = (int)(mu - (unsigned)b * hw);
      const int py = rem / a.Wq, px = rem - py * a.Wq;
      const int iy0 = a.transposed ? py + a.cy : py * a.stride - a.pad;
      const int ix0 = a.transposed ? px + a.cx : px * a.stride - a.pad;
      roff[i] = ((b * a.Hs + iy0) * a.Ws + ix0) * a.Cs + 4 * q;
      uint32_t mk = 0u;
      for (int t = 0; t < ntap; ++t) {
        const int tr = t / a.nS, ts = t - tr * a.nS;
        const int iy = iy0 + sgn * tr * a.dil, ix = ix0 + sgn * ts * a.dil;
        if (iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws) mk |= (1u << t);
      }
      rmask[i] = mk;
    }
  }
  // ---- hoisted B state ---------------------------------------------------------------------
  //   forward : tile [k][n]: n4 = tid&15, k = (tid>>4)+16*i ; element (tap*Cs + c0 + k)*Cd + n0 + 4*n4
  //   bwd-data: tile [n][k]: k4 = tid&7,  n = (tid>>3)+32*i ; element (tap*Cd + n0 + n)*Cs + c0 + 4*k4
  int boff[2];
  bool bok[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    if (!a.transposed) {
      const int n = n0 + 4 * (tid & 15), k = (tid >> 4) + 16 * i;
      boff[i] = k * a.Cd + n; bok[i] = n < a.Cd;
    } else {
      const int n = n0 + (tid >> 3) + 32 * i, k = 4 * (tid & 7);
      boff[i] = n * a.Cs + k; bok[i] = n < a.Cd;
    }
  }
  const int nchunk = (a.Cs + BK - 1) / BK;
  const int nstage = ntap * nchunk;
  const bool ktail = (a.Cs % BK) != 0;

  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.w, a.w_bytes);
  f32x4 ra[NR], rb[2];
  auto load_stage = [&](int st) {
    const int tapq = st / nchunk, c0 = (st - tapq * nchunk) * BK;
    const int tr = tapq / a.nS, ts = tapq - tr * a.nS;
    const int tap = (a.r0 + a.tstep * tr) * a.S + (a.s0 + a.tstep * ts);   // filter tap for the weights
    const int aoff = sgn * a.dil * (tr * a.Ws + ts) * a.Cs + c0;  // uniform
    const bool aok = !ktail || (c0 + 4 * q < a.Cs);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = ((rmask[i] >> tapq) & 1u) && aok;
      ra[i] = buf_load4(rsrc_a, ok ? (unsigned)(roff[i] + aoff) * 4u : OOB);
    }
    if (!a.transposed) {
      const int wbase = (tap * a.Cs + c0) * a.Cd;                 // uniform
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const bool ok = bok[i] && (!ktail || (c0 + (tid >> 4) + 16 * i < a.Cs));
        rb[i] = buf_load4(rsrc_w, ok ? (unsigned)(wbase + boff[i]) * 4u : OOB);
      }
    } else {
      const int wbase = tap * a.Cd * a.Cs + c0;                   // uniform
      const bool kok = !ktail || (c0 + 4 * (tid & 7) < a.Cs);
#pragma unroll
      for (int i = 0; i < 2; ++i)
        rb[i] = buf_load4(rsrc_w, (bok[i] && kok) ? (unsigned)(wbase + boff[i]) * 4u : OOB);
    }
  };
  auto store_stage = [&](int buf) {
    float* A = As[buf];
    float* Bm = Bs[buf];
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      float* p = &A[(arow + 32 * i) * A_LD + 4 * q];
      p[0] = ra[i].x; p[1] = ra[i].y; p[2] = ra[i].z; p[3] = ra[i].w;
    }
    if (!a.transposed) {
#pragma unroll
      for (int i = 0; i < 2; ++i)
        *reinterpret_cast<f32x4*>(&Bm[((tid >> 4) + 16 * i) * BN + 4 * (tid & 15)]) = rb[i];
    } else {
#pragma unroll
      for (int i = 0; i < 2; ++i) {
        const int n = (tid >> 3) + 32 * i, k = 4 * (tid & 7);
        Bm[(k + 0) * BN + n] = rb[i].x;
        Bm[(k + 1) * BN + n] = rb[i].y;
        Bm[(k + 2) * BN + n] = rb[i].z;
        Bm[(k + 3) * BN + n] = rb[i].w;
      }
    }
  };

  f32x16 acc[MT][2];
#pragma unroll
  for (int i = 0; i < MT; ++i) { acc[i][0] = f32x16{0}; acc[i][1] = f32x16{0}; }
  const int l31 = lane & 31, lh = lane >> 5;

  // Fragments are fetched a quarter-stage (4 k-pairs = 8*MT MFMAs) ahead of the MFMAs that use them,
  // so the LDS latency sits under the previous quarter's matrix work instead of in front of each pair.
  constexpr int QK = 4;                           // k-pairs per quarter
  struct Frag { float a[MT][QK]; float b0[QK], b1[QK]; };
  auto load_frag = [&](Frag& f, int buf, int qtr) {
    const float* Ap = &As[buf][(wave * 32 * MT + l31) * A_LD + lh + 2 * QK * qtr];
    const float* Bp = &Bs[buf][(lh + 2 * QK * qtr) * BN + l31];
#pragma unroll
    for (int kk = 0; kk < QK; ++kk) {
      f.b0[kk] = Bp[2 * kk * BN];
      f.b1[kk] = Bp[2 * kk * BN + 32];
#pragma unroll
      for (int i = 0; i < MT; ++i) f.a[i][kk] = Ap[i * 32 * A_LD + 2 * kk];
    }
  };
  auto mma_frag = [&](const Frag& f) {
#pragma unroll
    for (int kk = 0; kk < QK; ++kk)
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        acc[i][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][kk], f.b0[kk], acc[i][0], 0, 0, 0);
        acc[i][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f.a[i][kk], f.b1[kk], acc[i][1], 0, 0, 0);
      }
  };

  if (nstage > 0) {
    load_stage(0);
    store_stage(0);
  }
  if (nstage > 1) load_stage(1);
  __syncthreads();
  Frag f0, f1;
  for (int st = 0; st < nstage; ++st) {
    const int buf = st & 1;
    // sched_barrier(0) pins the order: hipcc otherwise sinks every ds_read next to its MFMA and
    // waits lgkmcnt(0) in front of each pair (measured: 62 % MFMA-pipe utilisation).
    load_frag(f0, buf, 0);
    load_frag(f1, buf, 1);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    __builtin_amdgcn_sched_barrier(0);
    load_frag(f0, buf, 2);
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f1);
    __builtin_amdgcn_sched_barrier(0);
    load_frag(f1, buf, 3);
    if (st + 1 < nstage) {
      store_stage(buf ^ 1);                       // waits for the loads issued one stage ago
      if (st + 2 < nstage) load_stage(st + 2);    // lands under the next stage's first half
    }
    __builtin_amdgcn_sched_barrier(0);
    mma_frag(f0);
    mma_frag(f1);
    __syncthreads();
  }

  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};     // BatchNorm statistics of this lane's 2 columns
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const int n = n0 + 32 * t + l31;
      if (n >= a.Cd) continue;
      const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const long m = m0 + wave * 32 * MT + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (m >= M) continue;
        float v = acc[i][t][r] + bv;
        long e = m * a.Cd + n;
        if (a.os > 1) {                             // strided sub-grid -> scattered destination pixel
          const int b = (int)(m / ((long)a.Hq * a.Wq));
          const int rem = (int)(m - (long)b * a.Hq * a.Wq);
          const int py = rem / a.Wq, px = rem - py * a.Wq;
          e = (((long)b * a.Hd + py * a.os + a.oy0) * a.Wd + px * a.os + a.ox0) * a.Cd + n;
        }
        if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
        a.dst[e] = v;
        csum[t] += v; csq[t] += v * v;
      }
    }
  if (a.stats != nullptr) {
    // lanes l and l+32 hold the same columns (rows +4): combine, then the 4 waves through LDS
    __syncthreads();                                  // all MFMA-stage LDS reads are finished
    float* red = &As[0][0];                           // [4 waves][64 cols][2]
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float s1 = csum[t] + __shfl_xor(csum[t], 32, 64), s2 = csq[t] + __shfl_xor(csq[t], 32, 64);
      if (lh == 0) { red[(wave * 64 + 32 * t + l31) * 2] = s1; red[(wave * 64 + 32 * t + l31) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < 128) {
      const int col = tid >> 1, which = tid & 1;
      const float v = red[(0 * 64 + col) * 2 + which] + red[(1 * 64 + col) * 2 + which] + red[(2 * 64 + col) * 2 + which] +
                      red[(3 * 64 + col) * 2 + which];
      const int n = n0 + col;
      if (n < a.Cd) a.stats[((long)(wg / a.ny) * a.Cd + n) * 2 + which] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// Split-precision variant of the affine kernel, templated over the operand split of conv_split.h: every fp32 operand is
// split on the fly into 16-bit planes while it is written to LDS and each fp32 product is the sum of the plane products that
// matter (fp32 accumulation):
//   PrecX3  ("bf16x3"): x = x1 + x2 + x3, six bf16 MFMAs (x1y3 + x2y2 + x3y1 + x1y2 + x2y1 + x1y1; the dropped terms are
//                       <= 2^-24 relative).  v_mfma_f32_32x32x16_bf16 runs at 16x the fp32-MFMA rate, so six of them cost
//                       3/8 of the eight v_mfma_f32_32x32x2_f32 they replace.
//   PrecF16 ("f16x2") : x * 2^s = h1 + h2, three fp16 MFMAs; s comes from the maxima of the LDS stage (source tile and weight
//                       tile separately), the accumulators carry the running exponent.
// Same tiling / addressing / epilogue as conv_igemm_affine_kernel; LDS holds NPL planes per operand, rows padded to 80 B
// so the ds_read_b128 fragment reads are conflict-free.
// ------------------------------------------------------------------------------------------
constexpr int XLD = 40;   // 16-bit elements per LDS row: 32 k + 8 pad (80 bytes)

template <class P>
__global__ __launch_bounds__(256) void conv_igemm_split_kernel(AffArgs a) {
  constexpr int TM = 128, NR = 4, NPL = P::NPL;
  typedef typename P::T T;
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  __shared__ __attribute__((aligned(16))) T Ap[NPL][TM * XLD];
  __shared__ __attribute__((aligned(16))) T Bp[NPL][BN * XLD];
  __shared__ unsigned amax_cell[2][2];      // [stage parity][source, weights]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const long M = (long)a.B * a.Hq * a.Wq;
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
  const long m0 = (long)(wg / a.ny) * TM;
  const int n0 = (wg % a.ny) * BN;
  const int q = tid & 7, arow = tid >> 3;
  const int sgn = a.transposed ? -1 : 1;
  const int ntap = a.nR * a.nS;
  const bool pointwise = ntap == 1 && a.Hq == a.Hs && a.Wq == a.Ws && a.os == 1 &&
                         (a.transposed ? (a.cy == 0 && a.cx == 0) : (a.stride == 1 && a.pad == 0));

  int roff[NR];
  uint32_t rmask[NR];
#pragma unroll
  for (int i = 0; i < NR; ++i) {
    const long m = m0 + arow + 32 * i;
    roff[i] = 0; rmask[i] = 0u;
    if (m < M && pointwise) {
      // single tap landing on the pixel itself (1x1 stride-1 convs, forward and bwd-data): source row = loop row, no (b, y, x)
      // decomposition -- the two integer divisions per row were ~30 % of this kernel's VALU instructions on 64->256 @ 80x80
      roff[i] = (int)m * a.Cs + 4 * q;
      rmask[i] = 1u;
    } else if (m < M) {
      const unsigned mu = (unsigned)m, hw = (unsigned)(a.Hq * a.Wq);       // M < 2^30: the source is below 4 GB (aligned_ok)
      const int b = (int)(mu / hw);
      const int rem = (int)(mu - (unsigned)b * hw);
      const int py = rem / a.Wq, px = rem - py * a.Wq;
      const int iy0 = a.transposed ? py + a.cy : py * a.stride - a.pad;
      const int ix0 = a.transposed ? px + a.cx : px * a.stride - a.pad;
      roff[i] = ((b * a.Hs + iy0) * a.Ws + ix0) * a.Cs + 4 * q;
      uint32_t mk = 0u;
      for (int t = 0; t < ntap; ++t) {
        const int tr = t / a.nS, ts = t - tr * a.nS;
        const int iy = iy0 + sgn * tr * a.dil, ix = ix0 + sgn * ts * a.dil;
        if (iy >= 0 && iy < a.Hs && ix >= 0 && ix < a.Ws) mk |= (1u << t);
      }
      rmask[i] = mk;
    }
  }
  // B loader: each thread fetches 8 consecutive k of one output column n
  //   forward : n = tid&63, k = 8*(tid>>6)+j, element (tap*Cs + c0 + k)*Cd + n0 + n       (8 dword loads)
  //   bwd-data: n = tid>>2, k = 8*(tid&3)+j,  element (tap*Cd + n0 + n)*Cs + c0 + k        (2 dwordx4 loads)
  const int bn = a.transposed ? (tid >> 2) : (tid & 63);
  const int bk0 = a.transposed ? 8 * (tid & 3) : 8 * (tid >> 6);
  const bool bnok = n0 + bn < a.Cd;
  const int nchunk = (a.Cs + BK - 1) / BK;
  const int nstage = ntap * nchunk;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.w, a.w_bytes);

  f32x4 ra[NR];
  float rb[8];
  auto load_stage = [&](int st) {
    const int tapq = st / nchunk, c0 = (st - tapq * nchunk) * BK;
    const int tr = tapq / a.nS, ts = tapq - tr * a.nS;
    const int tap = (a.r0 + a.tstep * tr) * a.S + (a.s0 + a.tstep * ts);
    const int aoff = sgn * a.dil * (tr * a.Ws + ts) * a.Cs + c0;
    const bool aok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      const bool ok = ((rmask[i] >> tapq) & 1u) && aok;
      ra[i] = buf_load4(rsrc_a, ok ? (unsigned)(roff[i] + aoff) * 4u : OOB);
    }
    if (!a.transposed) {
      const int wbase = (tap * a.Cs + c0 + bk0) * a.Cd + n0 + bn;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const bool ok = bnok && (c0 + bk0 + j < a.Cs);
        rb[j] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsrc_w, ok ? (int)((unsigned)(wbase + j * a.Cd) * 4u) : (int)OOB, 0, 0));
      }
    } else {
      const int wbase = (tap * a.Cd + n0 + bn) * a.Cs + c0 + bk0;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const bool ok = bnok && (c0 + bk0 + 4 * h < a.Cs);
        const f32x4 v = buf_load4(rsrc_w, ok ? (unsigned)(wbase + 4 * h) * 4u : OOB);
        rb[4 * h] = v.x; rb[4 * h + 1] = v.y; rb[4 * h + 2] = v.z; rb[4 * h + 3] = v.w;
      }
    }
  };
  auto store_stage = [&](float sa, float sb) {
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      X4 p[NPL];
      P::split4(P::SCALED ? ra[i] * sa : ra[i], p);
      const int o = (arow + 32 * i) * XLD + 4 * q;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ap[pl][o]) = p[pl];
    }
    X8 w[NPL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const f32x4 v = {rb[4 * h], rb[4 * h + 1], rb[4 * h + 2], rb[4 * h + 3]};
      X4 p[NPL];
      P::split4(P::SCALED ? v * sb : v, p);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
#pragma unroll
        for (int e = 0; e < 4; ++e) w[pl][4 * h + e] = p[pl][e];
    }
    const int o = bn * XLD + bk0;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X8*>(&Bp[pl][o]) = w[pl];
  };

  f32x16 acc0 = {0}, acc1 = {0};
  const int l31 = lane & 31, lh = lane >> 5;
  const int aidx = (wave * 32 + l31) * XLD + 8 * lh;     // + 16*s
  const int bidx = l31 * XLD + 8 * lh;                   // + 32*t*XLD + 16*s

  // Scaled planes (f16x2): every stage's operands are scaled by powers of two taken from the stage's own maxima (ea, eb) and
  // the accumulators carry the running exponent E >= ea + eb -- the scheme of conv_wgrad.hip, per stage instead of per patch.
  int E = 2 * fs_split::EMIN - 1, par = 0;
  if (P::SCALED) {
    if (tid < 4) amax_cell[tid >> 1][tid & 1] = 0u;
    __syncthreads();
  }
  if (nstage > 0) load_stage(0);
  for (int st = 0; st < nstage; ++st) {
    float sa = 1.f, sb = 1.f;
    if (P::SCALED) {
      float ma = 0.f, mb = 0.f;
#pragma unroll
      for (int i = 0; i < NR; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) ma = fmaxf(ma, fabsf(ra[i][e]));
#pragma unroll
      for (int j = 0; j < 8; ++j) mb = fmaxf(mb, fabsf(rb[j]));
      ma = wave_max(ma); mb = wave_max(mb);
      if (lane == 0) { atomicMax(&amax_cell[par][0], __builtin_bit_cast(unsigned, ma)); atomicMax(&amax_cell[par][1], __builtin_bit_cast(unsigned, mb)); }
    }
    __syncthreads();
    if (P::SCALED) {
      const int ea = __builtin_amdgcn_readfirstlane(fs_split::exponent_of_bits(amax_cell[par][0]));
      const int eb = __builtin_amdgcn_readfirstlane(fs_split::exponent_of_bits(amax_cell[par][1]));
      if (ea + eb > E) {
        const float f = fs_split::pow2f(E - ea - eb);
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc0[r] *= f; acc1[r] *= f; }
        E = ea + eb;
      }
      sa = fs_split::pow2f(14 - (E - eb)); sb = fs_split::pow2f(14 - eb);
      par ^= 1;
      if (tid < 2) amax_cell[par][tid] = 0u;
    }
    store_stage(sa, sb);
    __syncthreads();
    if (st + 1 < nstage) load_stage(st + 1);
    X8 fa[2][NPL], fb[2][2][NPL];
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2)
#pragma unroll
      for (int p3 = 0; p3 < NPL; ++p3) {
        fa[s2][p3] = *reinterpret_cast<const X8*>(&Ap[p3][aidx + 16 * s2]);
        fb[s2][0][p3] = *reinterpret_cast<const X8*>(&Bp[p3][bidx + 16 * s2]);
        fb[s2][1][p3] = *reinterpret_cast<const X8*>(&Bp[p3][bidx + 32 * XLD + 16 * s2]);
      }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      // smallest cross terms first
#pragma unroll
      for (int t = 0; t < P::NTERM; ++t) {
        acc0 = P::mfma(fa[s2][P::ta(t)], fb[s2][0][P::tb(t)], acc0);
        acc1 = P::mfma(fa[s2][P::ta(t)], fb[s2][1][P::tb(t)], acc1);
      }
    }
    __builtin_amdgcn_sched_barrier(0);
  }

  const int Eo = E - 28;                                   // two factors: the combined exponent can leave the float range
  const float fo1 = P::SCALED ? fs_split::pow2f(Eo / 2) : 1.f, fo2 = P::SCALED ? fs_split::pow2f(Eo - Eo / 2) : 1.f;
  float csum[2] = {0.f, 0.f}, csq[2] = {0.f, 0.f};
  // Epilogue with the row arithmetic hoisted out of the column loop and 32-bit buffer stores when dX / Y is below 4 GB: at
  // K = 64 (the 1x1 bottleneck convs) the epilogue's VALU work, not the MFMAs, was the larger share of the kernel.
  float bv[2];
  bool nok[2];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int n = n0 + 32 * t + l31;
    nok[t] = n < a.Cd;
    bv[t] = (a.bias != nullptr && nok[t]) ? a.bias[n] : 0.f;
  }
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
  const long HWq = (long)a.Hq * a.Wq;
#pragma unroll
  for (int r = 0; r < 16; ++r) {
    const long m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
    if (m >= M) continue;
    long erow = m * a.Cd;
    if (a.os > 1) {
      const int b = (int)(m / HWq);
      const int rem = (int)(m - (long)b * HWq);
      const int py = rem / a.Wq, px = rem - py * a.Wq;
      erow = (((long)b * a.Hd + py * a.os + a.oy0) * a.Wd + px * a.os + a.ox0) * a.Cd;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (!nok[t]) continue;
      float v = (P::SCALED ? (t == 0 ? acc0[r] : acc1[r]) * fo1 * fo2 : (t == 0 ? acc0[r] : acc1[r])) + bv[t];
      const long e = erow + n0 + 32 * t + l31;
      if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
      if (a.dst_bytes != 0u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(int, v), rsrc_d, (int)((unsigned)e * 4u), 0, 0);
      else a.dst[e] = v;
      csum[t] += v; csq[t] += v * v;
    }
  }
  if (a.stats != nullptr) {
    __syncthreads();
    float* red = reinterpret_cast<float*>(&Ap[0][0]);
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      const float s1 = csum[t] + __shfl_xor(csum[t], 32, 64), s2 = csq[t] + __shfl_xor(csq[t], 32, 64);
      if (lh == 0) { red[(wave * 64 + 32 * t + l31) * 2] = s1; red[(wave * 64 + 32 * t + l31) * 2 + 1] = s2; }
    }
    __syncthreads();
    if (tid < 128) {
      const int col = tid >> 1, which = tid & 1;
      const float v = red[(0 * 64 + col) * 2 + which] + red[(1 * 64 + col) * 2 + which] + red[(2 * 64 + col) * 2 + which] +
                      red[(3 * 64 + col) * 2 + which];
      const int n = n0 + col;
      if (n < a.Cd) a.stats[((long)(wg / a.ny) * a.Cd + n) * 2 + which] = v;
    }
  }
}

// ------------------------------------------------------------------------------------------
// bwd-weight: per tap GEMM  dW[tap][ci][co] += sum_{pix in split} X[pix+tap][ci] * dY[pix][co]
// tile 64(ci) x 64(co), K-step 32 pixels, one 32x32 accumulator per wave, split-K over
// blockIdx.z with fp32 atomic accumulation into a zeroed dW.
// ------------------------------------------------------------------------------------------
struct WgradArgs {
  const float* x;    // (B,H,W,Cin)
  const float* dy;   // (B,Ho,Wo,Cout)
  float* dw;         // [R][S][Cin][Cout], zero-initialised
  int B, H, W, Cin, Ho, Wo, Cout;
  int R, S, stride, pad, dil;
  int pix_per_split;
  int tiles, ntap, nsplit;
  FsPart part;       // deterministic mode (conv_kernels.h): slab `split` takes this workgroup's tile
};

template <bool VEC>
__global__ __launch_bounds__(256) void conv_wgrad_kernel(WgradArgs a) {
  __shared__ float As[BK * 64];   // [pix][ci]
  __shared__ float Bs[BK * 64];   // [pix][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (a.Cout + 63) / 64;
  // Plain dispatch order (tile fastest, then tap, then pixel split).  An XCD-contiguous remap of the
  // splits was measured 8-10 % SLOWER here (72 vs 79 TF on 64->64@80^2), so it is not applied.
  const int wg = blockIdx.x;
  const int split = wg / (a.tiles * a.ntap);
  const int rest = wg - split * (a.tiles * a.ntap);
  const int tap = rest / a.tiles, tile = rest - tap * a.tiles;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int ci0 = tm * 64, co0 = tn * 64;
  const int tr = tap / a.S, ts = tap - tr * a.S;
  const long P = (long)a.B * a.Ho * a.Wo;
  const long p_begin = (long)split * a.pix_per_split;
  const long p_end = (p_begin + a.pix_per_split < P) ? p_begin + a.pix_per_split : P;

  const int c4 = 4 * (tid & 15), prow = tid >> 4;   // rows prow + 16*i, i<2
  f32x4 ra[2], rb[2];

  auto load_stage = [&](long p0) {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long p = p0 + prow + 16 * i;
      f32x4 va = {0.f, 0.f, 0.f, 0.f}, vb = {0.f, 0.f, 0.f, 0.f};
      if (p < p_end) {
        const int b = (int)(p / ((long)a.Ho * a.Wo));
        const int rem = (int)(p - (long)b * a.Ho * a.Wo);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        const int iy = oy * a.stride - a.pad + tr * a.dil, ix = ox * a.stride - a.pad + ts * a.dil;
        if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) {
          const int c = ci0 + c4;
          const float* q = a.x + (((long)b * a.H + iy) * a.W + ix) * a.Cin + c;
          if (VEC) {
            if (c < a.Cin) va = *reinterpret_cast<const f32x4*>(q);
          } else {
            if (c + 0 < a.Cin) va.x = q[0];
            if (c + 1 < a.Cin) va.y = q[1];
            if (c + 2 < a.Cin) va.z = q[2];
            if (c + 3 < a.Cin) va.w = q[3];
          }
        }
        {
          const int c = co0 + c4;
          const float* q = a.dy + p * a.Cout + c;
          if (VEC) {
            if (c < a.Cout) vb = *reinterpret_cast<const f32x4*>(q);
          } else {
            if (c + 0 < a.Cout) vb.x = q[0];
            if (c + 1 < a.Cout) vb.y = q[1];
            if (c + 2 < a.Cout) vb.z = q[2];
            if (c + 3 < a.Cout) vb.w = q[3];
          }
        }
      }
      ra[i] = va; rb[i] = vb;
    }
  };
  auto store_stage = [&]() {
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&As[(prow + 16 * i) * 64 + c4]) = ra[i];
      *reinterpret_cast<f32x4*>(&Bs[(prow + 16 * i) * 64 + c4]) = rb[i];
    }
  };

  f32x16 acc = {0};
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;
  const float* Ap = &As[lh * 64 + 32 * wm + l31];
  const float* Bp = &Bs[lh * 64 + 32 * wn + l31];

  if (p_begin < p_end) load_stage(p_begin);
  for (long p0 = p_begin; p0 < p_end; p0 += BK) {
    __syncthreads();
    store_stage();
    __syncthreads();
    if (p0 + BK < p_end) load_stage(p0 + BK);
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk)
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(Ap[2 * kk * 64], Bp[2 * kk * 64], acc, 0, 0, 0);
  }
  const int co = co0 + 32 * wn + l31;
  if (co < a.Cout) {
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
      if (ci < a.Cin) fs_wgrad_out(a.dw, a.part, split, ((long)tap * a.Cin + ci) * a.Cout + co, acc[r]);
    }
  }
}

// ------------------------------------------------------------------------------------------
// bwd-weight, 3x3 filters, TG taps per workgroup (TG = 3: one filter row, TG = 9: whole filter).
// The per-tap kernel above re-reads X and dY once per tap through L2 (measured 905 MB of fabric
// traffic per launch against 210 MB algorithmic on 64->64@80^2); here a workgroup stages the dY chunk
// once per pixel chunk, keeps its MFMA B-fragments in registers for all its taps, and only the shifted
// X tile changes from tap to tap (neighbouring taps hit the same lines in L1/L2).
//   tile 64(ci) x 64(co) per tap, one 32x32 accumulator per wave per tap, 32-pixel chunks.
// ------------------------------------------------------------------------------------------
template <int TG>
__global__ __launch_bounds__(256) void conv_wgrad_taps_kernel(WgradArgs a) {
  __shared__ float Xs[2][BK * 64];   // [pix][ci], double-buffered across taps
  __shared__ float Ys[BK * 64];      // [pix][co]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int tiles_n = (a.Cout + 63) / 64;
  constexpr int NG = 9 / TG;                       // tap groups
  int wg = blockIdx.x;
  const int tile = wg % a.tiles; wg /= a.tiles;
  const int grp = wg % NG;
  const int split = wg / NG;
  const int tm = tile / tiles_n, tn = tile - tm * tiles_n;
  const int ci0 = tm * 64, co0 = tn * 64;
  const long P = (long)a.B * a.Ho * a.Wo;
  const long p_begin = (long)split * a.pix_per_split;
  const long p_end = (p_begin + a.pix_per_split < P) ? p_begin + a.pix_per_split : P;

  const __amdgpu_buffer_rsrc_t rx = make_rsrc(a.x, (unsigned)((size_t)a.B * a.H * a.W * a.Cin * 4));
  const __amdgpu_buffer_rsrc_t ry = make_rsrc(a.dy, (unsigned)((size_t)P * a.Cout * 4));
  const int c4 = 4 * (tid & 15), prow = tid >> 4;  // rows prow + 16*i
  const bool xc_ok = ci0 + c4 < a.Cin, yc_ok = co0 + c4 < a.Cout;
  const int l31 = lane & 31, lh = lane >> 5;
  const int wm = wave >> 1, wn = wave & 1;

  f32x16 acc[TG];
#pragma unroll
  for (int t = 0; t < TG; ++t) acc[t] = f32x16{0};

  for (long p0 = p_begin; p0 < p_end; p0 += BK) {
    // per-row state for this chunk: element offset of tap (0,0) and 9-bit tap validity
    int xoff[2];
    uint32_t xmask[2];
    unsigned yoff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const long p = p0 + prow + 16 * i;
      xoff[i] = 0; xmask[i] = 0u; yoff[i] = OOB;
      if (p < p_end) {
        const int b = (int)(p / ((long)a.Ho * a.Wo));
        const int rem = (int)(p - (long)b * a.Ho * a.Wo);
        const int oy = rem / a.Wo, ox = rem - oy * a.Wo;
        const int iy0 = oy * a.stride - a.pad, ix0 = ox * a.stride - a.pad;
        xoff[i] = ((b * a.H + iy0) * a.W + ix0) * a.Cin + ci0 + c4;
        uint32_t mk = 0u;
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int iy = iy0 + (t / 3) * a.dil, ix = ix0 + (t % 3) * a.dil;
          if (iy >= 0 && iy < a.H && ix >= 0 && ix < a.W) mk |= 1u << t;
        }
        xmask[i] = xc_ok ? mk : 0u;
        if (yc_ok) yoff[i] = (unsigned)(p * a.Cout + co0 + c4) * 4u;
      }
    }
    auto load_x = [&](int tap, f32x4* r) {
      const int toff = a.dil * ((tap / 3) * a.W + (tap % 3)) * a.Cin;
#pragma unroll
      for (int i = 0; i < 2; ++i) r[i] = buf_load4(rx, ((xmask[i] >> tap) & 1u) ? (unsigned)(xoff[i] + toff) * 4u : OOB);
    };
    f32x4 rx0[2], ry0[2];
    load_x(grp * TG, rx0);
#pragma unroll
    for (int i = 0; i < 2; ++i) ry0[i] = buf_load4(ry, yoff[i]);
    __syncthreads();                                 // previous chunk's LDS reads are done
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      *reinterpret_cast<f32x4*>(&Xs[0][(prow + 16 * i) * 64 + c4]) = rx0[i];
      *reinterpret_cast<f32x4*>(&Ys[(prow + 16 * i) * 64 + c4]) = ry0[i];
    }
    __syncthreads();
    float fb[BK / 2];
#pragma unroll
    for (int kk = 0; kk < BK / 2; ++kk) fb[kk] = Ys[(2 * kk + lh) * 64 + 32 * wn + l31];
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      f32x4 rn[2];
      if (t + 1 < TG) load_x(grp * TG + t + 1, rn);
      const float* Xp = &Xs[t & 1][lh * 64 + 32 * wm + l31];
      float fa[BK / 2];
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) fa[kk] = Xp[2 * kk * 64];
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int kk = 0; kk < BK / 2; ++kk) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(fa[kk], fb[kk], acc[t], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      if (t + 1 < TG) {
#pragma unroll
        for (int i = 0; i < 2; ++i) *reinterpret_cast<f32x4*>(&Xs[(t + 1) & 1][(prow + 16 * i) * 64 + c4]) = rn[i];
        __syncthreads();
      }
    }
  }
  const int co = co0 + 32 * wn + l31;
  if (co < a.Cout) {
#pragma unroll
    for (int t = 0; t < TG; ++t) {
      const int tap = grp * TG + t;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * lh;
        if (ci < a.Cin) fs_wgrad_out(a.dw, a.part, split, ((long)tap * a.Cin + ci) * a.Cout + co, acc[t][r]);
      }
    }
  }
}

// dst pixels (y, x) whose class (y % st, x % st) is in `classes` <- 0, all channels (16-byte stores, nothing read)
__global__ __launch_bounds__(256) void zero_classes_kernel(float* __restrict__ dst, long n4, int H, int W, int C4, int st, unsigned classes) {
  const f32x4 z = {0.f, 0.f, 0.f, 0.f};
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long)gridDim.x * 256) {
    const long pix = i / C4;
    const int x = (int)(pix % W), y = (int)((pix / W) % H);
    if ((classes >> ((y % st) * st + (x % st))) & 1u) reinterpret_cast<f32x4*>(dst)[i] = z;
  }
}

int launch_affine_one(AffArgs& a) {
  const long M = (long)a.B * a.Hq * a.Wq;
  a.nx = cdiv(M, 128);
  a.ny = cdiv(a.Cd, BN);
  // MT=2 (256-row tiles) needs 84 KB of LDS = one workgroup per CU and measured 20-25 % slower.
  if (g_conv_precision == 2)
    hipLaunchKernelGGL(conv_igemm_split_kernel<fs_split::PrecF16>, dim3(a.nx * a.ny), dim3(256), 0, a_stream, a);
  else if (g_conv_precision == 1)
    hipLaunchKernelGGL(conv_igemm_split_kernel<fs_split::PrecX3>, dim3(a.nx * a.ny), dim3(256), 0, a_stream, a);
  else
    hipLaunchKernelGGL(conv_igemm_affine_kernel<1>, dim3(a.nx * a.ny), dim3(256), 0, a_stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// The halo-tiled 3x3 kernel (conv_halo.hip) runs when the split-precision mode is on, the shape qualifies and the
// caller handed over enough scratch for the weight pack.
long halo_pack_bytes(int Cs, int Cd) { return fs_halo_pack_bytes(g_conv_precision, Cs, Cd); }
bool use_halo(const ConvArgs& c) {
  return g_conv_precision >= 1 && c.ws_ != nullptr && fs_halo_eligible(c.Hd, c.Wd, c.Cs, c.Cd, c.R, c.S, c.stride, c.pad, c.dil) &&
         c.Hs == c.Hd && c.Ws == c.Wd && c.ws_bytes_ >= halo_pack_bytes(c.Cs, c.Cd) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL &&      // 32-bit store offsets in the epilogue
         (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;        // 32-bit buffer offsets of the source (fs_halo_conv3x3 rejects larger)
}

// ... and its F(2,3) variant (conv_wino.hip) where the shape additionally allows the pair tiling.
bool use_wino(const ConvArgs& c) {
  return use_halo(c) && fs_wino_eligible(g_conv_precision, c.B, c.Hd, c.Wd, c.Cs, c.Cd) && c.ws_bytes_ >= fs_wino_pack_bytes(g_conv_precision, c.Cs, c.Cd);
}

// Everything else that is channel-aligned, undilated and has scratch goes to the tap-class kernel (conv_tapset.hip).
bool tapset_shape_ok(int Cs, int Cd, int R, int S, int stride, int dil) {
  const int cr = stride < R ? stride : R, cs = stride < S ? stride : S;
  // 1x1 filters have no tap reuse: measured slower than conv_igemm_split_kernel (43 vs 55 TF on 64->256 @ 80x80), not routed here
  // stride >= filter size: every tap class is a single tap, nothing is shared between outputs -- same case as 1x1
  // (960->512 3x3 stride 4 forward: 1.73 ms on conv_igemm_split_kernel, 2.37 ms here)
  if (stride >= R && stride >= S) return false;
  return dil == 1 && R * S > 1 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 16 && cr * cs <= 9 &&
         ((R + stride - 1) / stride) * ((S + stride - 1) / stride) <= 64;
}
long tapset_pack_bytes(int Cs, int Cd, int taps) { return fs_tapset_pack_bytes(g_conv_precision, Cs, Cd, taps); }
bool use_tapset(const ConvArgs& c) {
  return g_conv_precision >= 1 && c.ws_ != nullptr && tapset_shape_ok(c.Cs, c.Cd, c.R, c.S, c.stride, c.dil) &&
         c.ws_bytes_ >= tapset_pack_bytes(c.Cs, c.Cd, c.R * c.S) && (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL &&
         (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;        // the tap-class kernel addresses the source with 32-bit byte offsets too
}

// forward of 3x3 / stride 2 / pad 1 layers: the four input parity planes in one LDS refill per chunk (conv_s2fwd.hip)
static const bool g_s2fwd = FS_ENV_INT("FS_S2FWD", 1) != 0;      // kernel A/B builds only
bool use_s2fwd(const ConvArgs& c) {
  return g_s2fwd && g_conv_precision >= 1 && !c.transposed && c.ws_ != nullptr && c.bn_ == nullptr &&
         fs_s2fwd_eligible(c.Hs, c.Ws, c.Cs, c.Hd, c.Wd, c.Cd, c.R, c.S, c.stride, c.pad, c.dil) &&
         c.ws_bytes_ >= fs_s2fwd_pack_bytes(g_conv_precision, c.Cs, c.Cd) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}

// bwd-data of 3x3 / stride 2 / pad 1 layers: all four output parities in one launch (conv_s2bwd.hip)
static const bool g_s2bwd = FS_ENV_INT("FS_S2BWD", 1) != 0;      // kernel A/B builds only
bool use_s2bwd(const ConvArgs& c) {
  return g_s2bwd && g_conv_precision >= 1 && c.transposed && c.ws_ != nullptr && c.bias == nullptr &&
         fs_s2bwd_eligible(c.Hd, c.Wd, c.Cd, c.Hs, c.Ws, c.Cs, c.R, c.S, c.stride, c.pad, c.dil) &&
         c.ws_bytes_ >= fs_s2bwd_pack_bytes(g_conv_precision, c.Cd, c.Cs) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}

// 1x1 / stride-1 layers go to the GEMM kernel with pre-split weights (conv_pointwise.hip) when the caller handed over its scratch.
static const bool g_pointwise = FS_ENV_INT("FS_POINTWISE", 1) != 0;
bool use_pointwise(const ConvArgs& c) {
  return g_pointwise && g_conv_precision >= 1 && c.ws_ != nullptr && fs_pointwise_eligible(c.Cs, c.Cd, c.R, c.S, c.stride, c.pad, c.dil) &&
         c.Hs == c.Hd && c.Ws == c.Wd && c.ws_bytes_ >= fs_pointwise_pack_bytes(g_conv_precision, c.Cs, c.Cd) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}

// forward of stride >= filter layers (3x3 / stride 4, 1x1 / stride 4): the 1x1 GEMM kernel over gathered rows (conv_pointwise.hip)
static const bool g_pw_gather = FS_ENV_INT("FS_PW_GATHER", 1) != 0;      // kernel A/B builds only
bool use_pw_gather(const ConvArgs& c) {
  return g_pw_gather && g_conv_precision >= 1 && !c.transposed && c.ws_ != nullptr && c.bn_ == nullptr &&
         fs_pointwise_gather_eligible(c.Cs, c.Cd, c.R, c.S, c.stride, c.dil) &&
         c.ws_bytes_ >= fs_pointwise_pack_bytes(g_conv_precision, c.R * c.S * c.Cs, c.Cd) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}

// ... and their bwd-data: one 1x1 GEMM per tap, rows scattered to the tap's residue class of dX
bool use_pw_scatter(const ConvArgs& c) {
  return g_pw_gather && g_conv_precision >= 1 && c.transposed && c.ws_ != nullptr && c.bias == nullptr && c.bn_ == nullptr &&
         c.stride * c.stride <= 32 && fs_pointwise_scatter_eligible(c.Cd, c.Cs, c.R, c.S, c.stride, c.dil) &&
         c.ws_bytes_ >= fs_pointwise_pack_bytes(g_conv_precision, c.Cs, c.Cd) &&
         (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}

// The channel-aligned kernels (plain / halo / tap-class) address the SOURCE tensor with 32-bit BYTE offsets into a raw buffer
// resource (AffArgs::src_bytes), so the source must stay below 4 GB -- 2^30 elements, not 2^31; larger problems take the
// 64-bit-indexed conv_igemm_kernel.
bool aligned_ok(const ConvArgs& c) {
  return c.Cs % 4 == 0 && c.Cd % 4 == 0 && c.R * c.S <= 32 && (size_t)c.B * c.Hs * c.Ws * c.Cs * 4 < 4294967000UL;
}
int run_tapset(const FsTapsetProblem& p, hipStream_t stream) { return fs_tapset_conv(g_conv_precision, p, stream); }
FsTapsetProblem tapset_base(const ConvArgs& c) {
  FsTapsetProblem p{};
  p.src = c.src; p.w = c.w; p.bias = c.bias; p.dst = c.dst; p.stats = c.stats_; p.ws = c.ws_; p.w_amax = c.w_amax_;
  p.B = c.B; p.Hs = c.Hs; p.Ws = c.Ws; p.Cs = c.Cs; p.Hd = c.Hd; p.Wd = c.Wd; p.Cd = c.Cd;
  p.Cin = c.transposed ? c.Cd : c.Cs; p.Cout = c.transposed ? c.Cs : c.Cd; p.R = c.R; p.S = c.S;
  p.transposed = c.transposed;
  p.drop_scale = c.drop_scale; p.drop_thresh = c.drop_thresh; p.drop_key = c.drop_key;
  return p;
}
int launch_tapset_forward(const ConvArgs& c) {
  FsTapsetProblem p = tapset_base(c);
  p.Hq = c.Hd; p.Wq = c.Wd; p.os = 1; p.oy0 = 0; p.ox0 = 0; p.sm = c.stride;
  const int st = c.stride;
  p.ncls = 0;
  for (int r0 = 0; r0 < st && r0 < c.R; ++r0)
    for (int s0 = 0; s0 < st && s0 < c.S; ++s0)
      p.cls[p.ncls++] = FsTapClass{r0 - c.pad, s0 - c.pad, (c.R - r0 + st - 1) / st, (c.S - s0 + st - 1) / st, r0, st, s0, st};
  return run_tapset(p, c.stream_);
}

// bwd-data extras (BatchNorm-backward sums / addend) outside the 3x3 stride-1 family: the kernel launch_affine reaches for this problem is
// the 1x1 GEMM or the one-launch stride-2 kernel (round 5).  Same predicates, same order as the dispatch below.
bool bnsum_beyond_wino(const ConvArgs& c) {
  // forward: only the residual form of a 1x1 layer (addend, no sums, no inference epilogue) on the 1x1 GEMM kernel
  if (!c.transposed)
    return c.bn_ != nullptr && c.bn_->y == nullptr && c.bn_->ep_scale == nullptr && c.bn_->add_src != nullptr && c.stats_ == nullptr &&
           use_pointwise(c) && !use_wino(c) && !use_halo(c) && !use_s2fwd(c) && !use_tapset(c);
  if (use_s2fwd(c)) return false;
  if (c.stride == 1 && use_tapset(c) && !use_halo(c)) return false;
  if (use_wino(c) || use_halo(c)) return false;
  if (use_pointwise(c)) return true;
  if (use_pw_gather(c) || c.stride == 1) return false;
  return use_s2bwd(c);
}

int launch_affine(const ConvArgs& c, long M) {
  AffArgs a{c.src, c.w, c.bias, c.dst, c.B, c.Hs, c.Ws, c.Cs, c.Hd, c.Wd, c.Cd, c.R, c.S, c.stride, c.pad, c.dil, c.transposed,
            c.drop_scale, c.drop_thresh, c.drop_key,
            (unsigned)((size_t)c.B * c.Hs * c.Ws * c.Cs * 4), (unsigned)((size_t)c.R * c.S * c.Cs * c.Cd * 4),
            (size_t)c.B * c.Hd * c.Wd * c.Cd * 4 < 4294967000UL ? (unsigned)((size_t)c.B * c.Hd * c.Wd * c.Cd * 4) : 0u,
            c.Hd, c.Wd, 1, 0, 0, 0, 0, 1, c.R, c.S, c.pad, c.pad, 0, 0, c.stats_};
  a_stream = c.stream_;
  (void)M;
  if (use_s2fwd(c))
    return fs_s2fwd_conv(g_conv_precision, c.src, c.w, c.bias, c.dst, c.stats_, c.ws_, c.w_amax_, c.B, c.Hs, c.Ws, c.Cs, c.Hd, c.Wd, c.Cd,
                         c.drop_scale, c.drop_thresh, c.drop_key, c.stream_);
  if (!c.transposed && use_tapset(c) && !use_halo(c)) return launch_tapset_forward(c);
  if (c.transposed && c.stride == 1 && use_tapset(c) && !use_halo(c)) {
    FsTapsetProblem p = tapset_base(c);
    p.Hq = c.Hd; p.Wq = c.Wd; p.os = 1; p.oy0 = 0; p.ox0 = 0; p.sm = 1;
    p.ncls = 1;
    p.cls[0] = FsTapClass{c.pad - (c.R - 1), c.pad - (c.S - 1), c.R, c.S, c.R - 1, -1, c.S - 1, -1};
    return run_tapset(p, c.stream_);
  }
  if (use_wino(c))
    return fs_wino_conv3x3(g_conv_precision, c.src, c.w, c.bias, c.dst, c.stats_, c.ws_, c.w_amax_, c.B, c.Hd, c.Wd, c.Cs, c.Cd,
                           c.transposed ? c.Cd : c.Cs, c.transposed ? c.Cs : c.Cd, c.transposed, c.drop_scale, c.drop_thresh, c.drop_key, c.bn_, c.stream_);
  if (c.bn_ != nullptr && !bnsum_beyond_wino(c)) return FS_ERR_ARG;        // the caller asked for fused sums on a shape fs_conv2d_bwd_data_bnsum_slabs reported 0 for
  if (use_halo(c))
    return fs_halo_conv3x3(g_conv_precision, c.src, c.w, c.bias, c.dst, c.stats_, c.ws_, c.w_amax_, c.B, c.Hd, c.Wd, c.Cs, c.Cd,
                           c.transposed ? c.Cd : c.Cs, c.transposed ? c.Cs : c.Cd, c.transposed, c.drop_scale, c.drop_thresh, c.drop_key,
                           c.stream_);
  if (use_pointwise(c))
    return fs_pointwise_conv(g_conv_precision, c.src, c.w, c.bias, c.dst, c.stats_, c.ws_, c.w_amax_, (long)c.B * c.Hd * c.Wd, c.Cs, c.Cd,
                             c.transposed ? c.Cd : c.Cs, c.transposed ? c.Cs : c.Cd, c.transposed, c.drop_scale, c.drop_thresh, c.drop_key,
                             c.bn_, c.stream_);
  if (use_pw_gather(c))
    return fs_pointwise_gather_conv(g_conv_precision, c.src, c.w, c.bias, c.dst, c.stats_, c.ws_, c.w_amax_, c.B, c.Hs, c.Ws, c.Cs, c.Hd, c.Wd, c.Cd,
                                    c.R, c.S, c.stride, c.pad, c.drop_scale, c.drop_thresh, c.drop_key, c.stream_);
  if ((!c.transposed || c.stride == 1) && fs_ws_mode_tls != 0) return FS_ERR_ARG;      // the plain kernel has no weight pack
  if (!c.transposed || c.stride == 1) return launch_affine_one(a);
  if (use_s2bwd(c))
    return fs_s2bwd_conv(g_conv_precision, c.src, c.w, c.dst, c.ws_, c.w_amax_, c.B, c.Hd, c.Wd, c.Cd, c.Hs, c.Ws, c.Cs, c.bn_, c.stats_, c.stream_);
  // stride>1 bwd-data: one dense sub-problem per output parity class (oy0, ox0).  dX pixel y receives
  // tap r iff (y + pad - r) % stride == 0, i.e. r = r0 + stride*t with r0 = (oy0 + pad) % stride, and then
  // reads dY row (y + pad - r)/stride = py + (oy0 + pad - r0)/stride - t.
  const int st = c.stride;
  if (fs_ws_mode_tls != 0) return FS_ERR_ARG;      // the parity sub-problems below re-pack ws one after the other: no pack outlives the call
  if (use_pw_scatter(c)) {
    unsigned empty = 0u;
    for (int oy0 = 0; oy0 < st; ++oy0)
      for (int ox0 = 0; ox0 < st; ++ox0)
        if ((oy0 + c.pad) % st >= c.R || (ox0 + c.pad) % st >= c.S) empty |= 1u << (oy0 * st + ox0);
    if (empty != 0u) {
      const long n4 = (long)c.B * c.Hd * c.Wd * (c.Cd / 4);
      long blocks = (n4 + 255) / 256;
      if (blocks > 65536) blocks = 65536;
      hipLaunchKernelGGL(zero_classes_kernel, dim3((unsigned)blocks), dim3(256), 0, c.stream_, c.dst, n4, c.Hd, c.Wd, c.Cd / 4, st, empty);
      FS_LAUNCH_CHECK();
    }
    return fs_pointwise_scatter_conv(g_conv_precision, c.src, c.w, c.dst, c.ws_, c.w_amax_, c.B, c.Hd, c.Wd, c.Cd, c.Hs, c.Ws, c.Cs, c.R, c.S,
                                     c.stride, c.pad, c.stream_);
  }
  // classes no tap reaches (stride > filter size: 7 of 16 for 3x3 stride 4, 15 of 16 for 1x1 stride 4) are zero-filled by one
  // store-only launch instead of one conv launch each (1x1 stride 4, 1.57 GB of dX: 1.13 -> 0.55 ms)
  const bool fill_ok = st * st <= 32 && c.Cd % 4 == 0 && c.bias == nullptr;
  unsigned empty_classes = 0u;
  for (int oy0 = 0; oy0 < st; ++oy0)
    for (int ox0 = 0; ox0 < st; ++ox0) {
      AffArgs b = a;
      b.os = st; b.oy0 = oy0; b.ox0 = ox0; b.tstep = st;
      b.Hq = (c.Hd - oy0 + st - 1) / st; b.Wq = (c.Wd - ox0 + st - 1) / st;
      b.r0 = (oy0 + c.pad) % st; b.s0 = (ox0 + c.pad) % st;
      b.nR = b.r0 < c.R ? (c.R - b.r0 + st - 1) / st : 0;
      b.nS = b.s0 < c.S ? (c.S - b.s0 + st - 1) / st : 0;
      if (b.nR == 0 || b.nS == 0) { b.nR = 0; b.nS = 0; }       // no tap reaches this class: zeros
      b.cy = (oy0 + c.pad - b.r0) / st; b.cx = (ox0 + c.pad - b.s0) / st;
      if (b.Hq <= 0 || b.Wq <= 0) continue;
      if (b.nR == 0 && fill_ok) { empty_classes |= 1u << (oy0 * st + ox0); continue; }
      if (b.nR * b.nS > 1 && use_tapset(c)) {       // single-tap sub-problems: no reuse, the plain kernel is faster
        // dY row of tap t is py + cy - t: in increasing source order tr = nR-1-t, filter row r0 + st*(nR-1-tr)
        FsTapsetProblem p = tapset_base(c);
        p.Hq = b.Hq; p.Wq = b.Wq; p.os = st; p.oy0 = oy0; p.ox0 = ox0; p.sm = 1;
        p.ncls = 1;
        p.cls[0] = FsTapClass{b.cy - (b.nR - 1), b.cx - (b.nS - 1), b.nR, b.nS, b.r0 + st * (b.nR - 1), -st, b.s0 + st * (b.nS - 1), -st};
        int e2 = run_tapset(p, c.stream_);
        if (e2 != FS_OK) return e2;
        continue;
      }
      int e = launch_affine_one(b);
      if (e != FS_OK) return e;
    }
  if (empty_classes != 0u) {
    const long n4 = (long)c.B * c.Hd * c.Wd * (c.Cd / 4);
    long blocks = (n4 + 255) / 256;
    if (blocks > 65536) blocks = 65536;
    hipLaunchKernelGGL(zero_classes_kernel, dim3((unsigned)blocks), dim3(256), 0, c.stream_, c.dst, n4, c.Hd, c.Wd, c.Cd / 4, st, empty_classes);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

}  // namespace

bool fs_deterministic() { return g_deterministic != 0; }

int fs_wgrad_reduce(float* part, int nslab, long n, float* dw, int accumulate, hipStream_t stream) {
  return fs_slab_reduce_inplace(part, nslab, n, dw, accumulate, stream);
}

// slabs one bwd-weight call can need in deterministic mode: an upper bound of the split counts chosen below and in conv_wgrad.hip
static long wgrad_slab_cap(int Cin, int Cout, int R, int S, int stride, int pad, int dil) {
  const long tiles = (long)cdiv(Cin, 64) * cdiv(Cout, 64);
  if (g_conv_precision >= 1 && fs_wgrad_split_eligible(Cin, Cout, R, S, stride, pad, dil)) {
    if ((R == 1 && S == 1 && stride == 1 && pad == 0) || (stride >= R && stride >= S))
      return 2048 / tiles + 2;      // linear_wgrad_kernel (plain or gathered rows): up to 1024 workgroups of 64 x 64 tiles, 512 of wider ones
    return 512 / tiles + 2;                                                      // class kernels: 512 workgroups; transform-domain kernel: 2 x 256
  }
  if (R == 3 && S == 3 && Cin % 4 == 0 && Cout % 4 == 0) return 1024 / tiles + 2;
  return 2048 / (tiles * R * S) + 2;
}

extern "C" {

// include/fovealseg.h: fs_set_deterministic / fs_get_deterministic (host-side switch, no launch)
int fs_set_deterministic(int on) {
  g_deterministic = on ? 1 : 0;
  return FS_OK;
}
int fs_get_deterministic(void) { return g_deterministic; }

// include/fovealseg.h: scratch fs_conv2d_bwd_weight needs for this layer (0 unless deterministic mode is on)
// Strided 3x3 layers (one round of <= 512 workgroups: ~50-85 us of patch rounds, then ~80 us of split-K atomics with nothing left to
// overlap them): partial tiles by plain stores into per-split slabs + one ordered reduce launch instead of the atomics, in the default
// mode too -- the machinery of deterministic mode, without the memset (the one-launch kernel writes every element of every slab it uses).
static const int g_wgrad_store = FS_ENV_INT("FS_WGRAD_STORE", 1);      // kernel A/B builds only: 0 atomics everywhere, 2 also the 3x3 stride-1 class kernel
static bool wgrad_store_route(int Cin, int Cout, int R, int S, int stride, int pad, int dil) {
  if (g_wgrad_store == 0 || g_conv_precision < 1 || R != 3 || S != 3 || !fs_wgrad_split_eligible(Cin, Cout, R, S, stride, pad, dil)) return false;
  if (g_conv_precision == 1 && fs_wgrad_gather_s2(Cin, R, S, stride) && g_wgrad_store != 3) return false;      // those run as gathered-row GEMMs (conv_wgrad.hip)
  return stride == 2 || stride == 3 || (g_wgrad_store == 2 && stride == 1 && pad == 1);
}
long fs_conv2d_bwd_weight_ws_bytes(int Cin, int Cout, int R, int S, int stride, int pad, int dil) {
  if (Cin <= 0 || Cout <= 0 || R <= 0 || S <= 0 || stride <= 0) return 0;
  if (!g_deterministic && !wgrad_store_route(Cin, Cout, R, S, stride, pad, dil)) return 0;
  return wgrad_slab_cap(Cin, Cout, R, S, stride, pad, dil) * (long)R * S * Cin * Cout * 4;
}
// include/fovealseg.h: scratch of fs_linear_bwd_weight_bias (dW slabs, then 4 bias slabs per dW slab)
long fs_linear_bwd_weight_bias_ws_bytes(int Cin, int Cout) {
  if (!g_deterministic || Cin <= 0 || Cout <= 0) return 0;
  const long cap = wgrad_slab_cap(Cin, Cout, 1, 1, 1, 0, 1);
  return cap * ((long)Cin * Cout + 4L * Cout) * 4;
}

// include/fovealseg.h: fs_set_conv_precision / fs_get_conv_precision (host-side switch, no launch)
int fs_set_conv_precision(int mode) {
  FS_REQUIRE(mode >= 0 && mode <= 2);
  g_conv_precision = mode;
  return FS_OK;
}
int fs_get_conv_precision(void) { return g_conv_precision; }

// include/fovealseg.h: fs_weight_amax_segments -- max|w| (float bits) of nparams tensors laid out in one arena, one launch.
int fs_weight_amax_segments(const float* arena, const long* offsets, const long* sizes, int nparams, unsigned* out, hipStream_t stream) {
  FS_REQUIRE(arena && offsets && sizes && out && nparams > 0 && nparams <= 65535);
  return fs_weight_amax_segments_impl(arena, offsets, sizes, nparams, out, stream);
}

// include/fovealseg.h: fs_conv2d_workspace_bytes -- scratch the conv entry points can use for this shape (0 = none).
// transposed = 0 for fs_conv2d_fwd / fs_conv2d_fwd_stats, 1 for fs_conv2d_bwd_data.
long fs_conv2d_workspace_bytes(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil, int transposed) {
  if (g_conv_precision < 1) return 0;
  const int Cs = transposed ? Cout : Cin, Cd = transposed ? Cin : Cout;
  long need = 0;
  if (H == Ho && W == Wo && fs_halo_eligible(H, W, Cs, Cd, R, S, stride, pad, dil)) {
    need = halo_pack_bytes(Cs, Cd);
    const long t = fs_wino_pack_bytes(g_conv_precision, Cs, Cd);       // the F(2,3) variant packs 4 components per filter row
    if (t > need) need = t;
  }
  if (tapset_shape_ok(Cs, Cd, R, S, stride, dil)) {
    const long t = tapset_pack_bytes(Cs, Cd, R * S);
    if (t > need) need = t;
  }
  if (g_pointwise && H == Ho && W == Wo && fs_pointwise_eligible(Cs, Cd, R, S, stride, pad, dil)) {
    const long t = fs_pointwise_pack_bytes(g_conv_precision, Cs, Cd);
    if (t > need) need = t;
  }
  if (g_s2fwd && !transposed && fs_s2fwd_eligible(H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil)) {
    const long t = fs_s2fwd_pack_bytes(g_conv_precision, Cin, Cout);
    if (t > need) need = t;
  }
  if (g_pw_gather && !transposed && fs_pointwise_gather_eligible(Cin, Cout, R, S, stride, dil)) {
    const long t = fs_pointwise_pack_bytes(g_conv_precision, R * S * Cin, Cout);
    if (t > need) need = t;
  }
  if (g_pw_gather && transposed && fs_pointwise_scatter_eligible(Cin, Cout, R, S, stride, dil)) {
    const long t = fs_pointwise_pack_bytes(g_conv_precision, Cout, Cin);
    if (t > need) need = t;
  }
  if (g_s2bwd && transposed && fs_s2bwd_eligible(H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil)) {
    const long t = fs_s2bwd_pack_bytes(g_conv_precision, Cin, Cout);
    if (t > need) need = t;
  }
  return need;
}

// include/fovealseg.h: fs_conv2d_kernel_choice -- which kernel family the conv entry points select for this problem under the current
// precision mode and `ws_bytes` of scratch (host-side predicate, no launch; the dispatch below uses the same functions).
// 0 = generic 64-bit-indexed kernel, 1 = plain aligned implicit GEMM, 2 = halo-tiled 3x3, 3 = tap-class kernel, 4 = 1x1 GEMM kernel,
// 5 = halo-tiled 3x3 with F(2,3) minimal filtering along the row, 6 = stride-2 bwd-data with the four output parities in one launch,
// 7 = stride-2 forward with the four input parity planes in one LDS refill, 8 = halo-tiled 3x3 with F(4,3) minimal filtering along the
// row (conv_wino4.hip: bf16x3, widths that are multiples of 4).
int fs_conv2d_kernel_choice(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                            int transposed, long ws_bytes) {
  ConvArgs c{nullptr, nullptr, nullptr, nullptr, B, transposed ? Ho : H, transposed ? Wo : W, transposed ? Cout : Cin,
             transposed ? H : Ho, transposed ? W : Wo, transposed ? Cin : Cout, R, S, stride, pad, dil, transposed, 1.f, 0u, 0u};
  c.ws_ = ws_bytes > 0 ? (void*)&c : nullptr;
  c.ws_bytes_ = ws_bytes;
  if (!aligned_ok(c)) return 0;
  if (use_s2fwd(c)) return 7;
  if (!transposed && use_tapset(c) && !use_halo(c)) return 3;
  if (transposed && stride == 1 && use_tapset(c) && !use_halo(c)) return 3;
  if (use_wino(c)) return fs_wino_takes_f43(g_conv_precision, c.B, c.Hd, c.Wd, c.Cs, c.Cd) ? 8 : 5;
  if (use_halo(c)) return 2;
  if (use_pointwise(c) || use_pw_gather(c)) return 4;
  if (use_s2bwd(c)) return 6;
  if (transposed && stride > 1 && use_tapset(c)) return 3;     // the multi-tap parity sub-problems
  return 1;
}

// include/fovealseg.h: 1 when the scratch of this problem is ONE weight pack that depends on (w, shape, precision mode) only, so a caller
// may keep it across calls (fs_conv2d_pack once per weight update, then FS_WS_RUN_ONLY calls); 0 = no pack, or one that does not outlive the call
int fs_conv2d_pack_persistent(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                              int transposed, long ws_bytes) {
  const int k = fs_conv2d_kernel_choice(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed, ws_bytes);
  if (k == 2 || k == 4 || k == 5 || k == 6 || k == 7 || k == 8) return 1;
  return (k == 3 && !(transposed && stride > 1)) ? 1 : 0;
}

// include/fovealseg.h: set how the calling thread's next conv calls treat ws (0 pack + run, 1 run only: ws holds the pack); returns the old mode
int fs_conv2d_ws_mode(int mode) {
  const int old = fs_ws_mode_tls;
  if (mode == 0 || mode == FS_WS_RUN_ONLY) fs_ws_mode_tls = mode;
  return old;
}

// include/fovealseg.h: run only the weight pack of the kernel fs_conv2d_fwd* (transposed = 0) / fs_conv2d_bwd_data* (1) select for this problem
int fs_conv2d_pack(const float* w, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                   int transposed, void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  FS_REQUIRE(w && ws && ws_bytes > 0 && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0 && dil >= 1);
  FS_REQUIRE(Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  FS_REQUIRE(fs_conv2d_pack_persistent(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, transposed, ws_bytes) == 1);
  // src / dst are never dereferenced: every family returns after its pack launch in this mode (the predicate above excludes the rest)
  const float* nowhere = reinterpret_cast<const float*>(ws);
  ConvArgs a = transposed ? ConvArgs{nowhere, w, nullptr, reinterpret_cast<float*>(ws), B, Ho, Wo, Cout, H, W, Cin, R, S, stride, pad, dil, 1, 1.f, 0u, 0u}
                          : ConvArgs{nowhere, w, nullptr, reinterpret_cast<float*>(ws), B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, 0u};
  a.stream_ = stream;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  const int old = fs_ws_mode_tls;
  fs_ws_mode_tls = FS_WS_PACK_ONLY;
  const int e = launch_affine(a, 0);
  fs_ws_mode_tls = old;
  return e;
}

// include/fovealseg.h: fs_conv2d_stats_slabs -- number of [Cout][2] partial-sum slabs fs_conv2d_fwd_stats writes for
// this shape when called with ws_bytes of scratch (depends on which kernel it selects).
int fs_conv2d_stats_slabs(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                          long ws_bytes) {
  // the SAME predicates, in the same order, as the dispatch of fs_conv2d_fwd_stats (launch_affine): a count derived from eligibility alone
  // would describe another kernel's slab layout whenever a dispatch condition (precision mode, 4 GB bounds, scratch size) fails
  ConvArgs c{nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, 0u};
  c.ws_ = ws_bytes > 0 ? (void*)&c : nullptr;
  c.ws_bytes_ = ws_bytes;
  if (aligned_ok(c)) {
    if (use_s2fwd(c)) return fs_s2fwd_slabs(B, Ho, Wo);
    if (use_tapset(c) && !use_halo(c)) return fs_tapset_slabs(B, Ho, Wo, (R + stride - 1) / stride, (S + stride - 1) / stride);
    if (use_wino(c)) return fs_wino_stats_slabs(g_conv_precision, B, Ho, Wo, Cin, Cout);
    if (use_halo(c)) return fs_halo_stats_slabs(B, Ho, Wo);
  }
  return cdiv((long)B * Ho * Wo, 128);      // the 1x1 GEMM kernels, the plain aligned kernel and the generic one: one slab per 128 rows
}

// include/fovealseg.h: fs_conv2d_fwd
int fs_conv2d_fwd(const float* x, const float* w, const float* bias, float* y, int B, int H, int W, int Cin,
                  int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key,
                  void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  FS_REQUIRE(x && w && y && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0);
  FS_REQUIRE(dil >= 1 && Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  FS_REQUIRE(drop_p >= 0.f && drop_p < 1.f);
  ConvArgs a{x, w, bias, y, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, drop_key};
  a.stream_ = stream;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  if (drop_p > 0.f) {
    a.drop_scale = 1.0f / (float)(1.0 - (double)drop_p);
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
  }
  const long M = (long)B * Ho * Wo;
  FS_REQUIRE(M * Cout < 4294967296L);
  if (aligned_ok(a)) return launch_affine(a, M);
  dim3 grid(cdiv(M, BM), cdiv(Cout, BN));
  if ((Cin % 4 == 0) && (Cout % 4 == 0))
    hipLaunchKernelGGL(conv_igemm_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(conv_igemm_kernel<false>, grid, dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: fs_conv2d_fwd_residual -- y = res + DropPath(Dropout(conv(x, w) + bias)) for a layer the 1x1 GEMM kernel runs
int fs_conv2d_fwd_residual_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                              long rows_per_sample, long ws_bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || rows_per_sample <= 0 || ((long)B * Ho * Wo) % rows_per_sample) return 0;
  ConvArgs a{nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, 0u};
  static unsigned char dummy;
  static const float one = 1.f;
  const FsBnSums bn{nullptr, nullptr, nullptr, nullptr, &one, nullptr, nullptr, nullptr, nullptr, 0, 1.f, 0u, 0u, 0};
  a.ws_ = ws_bytes > 0 ? &dummy : nullptr; a.ws_bytes_ = ws_bytes; a.bn_ = &bn;
  return (aligned_ok(a) && bnsum_beyond_wino(a) && rows_per_sample >= 128 && rows_per_sample < 2147483647L && (long)B * Ho * Wo * Cout < 4294967296L) ? 1 : 0;
}

int fs_conv2d_fwd_residual(const float* x, const float* w, const float* bias, const float* res, float* y, int B, int H, int W, int Cin, int Ho,
                           int Wo, int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key, float droppath_p,
                           uint32_t droppath_key, long rows_per_sample, void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  FS_REQUIRE(x && w && res && y && drop_p >= 0.f && drop_p < 1.f && droppath_p >= 0.f && droppath_p < 1.f);
  FS_REQUIRE(fs_conv2d_fwd_residual_ok(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, rows_per_sample, ws_bytes) == 1);
  ConvArgs a{x, w, bias, y, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, drop_key};
  a.stream_ = stream;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  if (drop_p > 0.f) {
    a.drop_scale = 1.0f / (float)(1.0 - (double)drop_p);
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
  }
  FsBnSums bn{nullptr, nullptr, nullptr, nullptr, res, nullptr, nullptr, nullptr, nullptr, 0, 1.f, 0u, droppath_key, (int)rows_per_sample};
  if (droppath_p > 0.f) {
    bn.dp_scale = 1.0f / (float)(1.0 - (double)droppath_p);
    bn.dp_thresh = (uint32_t)((double)droppath_p * 4294967296.0);
  }
  a.bn_ = &bn;
  return launch_affine(a, (long)B * Ho * Wo);
}

// include/fovealseg.h: 1 when fs_conv2d_fwd_affine_act can serve this shape (the F(2,3) kernels' row epilogue), else 0
int fs_conv2d_fwd_affine_act_ok(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                                long ws_bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dil != 1 || Cout % 4) return 0;
  ConvArgs a{nullptr, nullptr, nullptr, nullptr, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, 0u};
  static unsigned char dummy;
  a.ws_ = ws_bytes > 0 ? &dummy : nullptr; a.ws_bytes_ = ws_bytes;
  return (aligned_ok(a) && !(use_tapset(a) && !use_halo(a)) && use_wino(a)) ? 1 : 0;
}

// include/fovealseg.h: inference forward  z = act((conv(x, w) + bias) * scale[c] + shift[c] [+ res])  in one launch
int fs_conv2d_fwd_affine_act(const float* x, const float* w, const float* bias, const float* scale, const float* shift, const float* res,
                             float* z, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                             int act, void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  FS_REQUIRE(x && w && z && scale && shift && act >= 0 && act <= 2);
  FS_REQUIRE(fs_conv2d_fwd_affine_act_ok(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes) == 1);
  FS_REQUIRE(Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  ConvArgs a{x, w, bias, z, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, 0u};
  a.stream_ = stream;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  const FsBnSums ep{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, scale, shift, res, act};
  a.bn_ = &ep;
  return launch_affine(a, (long)B * Ho * Wo);
}

// include/fovealseg.h: fs_conv2d_fwd_stats -- forward conv that also emits per-workgroup BatchNorm partials.
// stats = [ceil(B*Ho*Wo/128)][Cout][2] floats.  Requires Cin%4==0 && Cout%4==0 (the affine kernel).
int fs_conv2d_fwd_stats(const float* x, const float* w, const float* bias, float* y, float* stats, int B, int H, int W, int Cin,
                        int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil, float drop_p, uint32_t drop_key,
                        void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  FS_REQUIRE(x && w && y && stats && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0);
  FS_REQUIRE(dil >= 1 && Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  FS_REQUIRE(drop_p >= 0.f && drop_p < 1.f);
  ConvArgs a{x, w, bias, y, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, 0, 1.f, 0u, drop_key};
  FS_REQUIRE(aligned_ok(a));
  a.stream_ = stream;
  a.stats_ = stats;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  if (drop_p > 0.f) {
    a.drop_scale = 1.0f / (float)(1.0 - (double)drop_p);
    a.drop_thresh = (uint32_t)((double)drop_p * 4294967296.0);
  }
  const long M = (long)B * Ho * Wo;
  FS_REQUIRE(M * Cout < 4294967296L);
  return launch_affine(a, M);
}

// include/fovealseg.h: fs_conv2d_bwd_data   (dX has the forward input's shape B,H,W,Cin)
static int conv2d_bwd_data_impl(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                                int Cout, int R, int S, int stride, int pad, int dil, void* ws, long ws_bytes, const unsigned* w_amax,
                                const FsBnSums* bn, float* slab, hipStream_t stream) {
  FS_REQUIRE(dy && w && dx && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0);
  FS_REQUIRE(dil >= 1 && Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  ConvArgs a{dy, w, nullptr, dx, B, Ho, Wo, Cout, H, W, Cin, R, S, stride, pad, dil, 1, 1.f, 0u, 0u};
  FS_REQUIRE(stride == 1 || dil == 1);
  const long M = (long)B * H * W;
  a.stream_ = stream;
  a.ws_ = ws; a.ws_bytes_ = ws_bytes; a.w_amax_ = w_amax;
  a.bn_ = bn; a.stats_ = slab;
  if (aligned_ok(a)) return launch_affine(a, M);
  FS_REQUIRE(bn == nullptr);
  dim3 grid(cdiv(M, BM), cdiv(Cin, BN));
  if ((Cin % 4 == 0) && (Cout % 4 == 0))
    hipLaunchKernelGGL(conv_igemm_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(conv_igemm_kernel<false>, grid, dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

int fs_conv2d_bwd_data(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo,
                       int Cout, int R, int S, int stride, int pad, int dil, void* ws, long ws_bytes, const unsigned* w_amax, hipStream_t stream) {
  return conv2d_bwd_data_impl(dy, w, dx, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws, ws_bytes, w_amax, nullptr, nullptr, stream);
}

// include/fovealseg.h: rows of the slab fs_conv2d_bwd_data_bnsum writes for this problem, 0 = this shape's kernel cannot form the sums
int fs_conv2d_bwd_data_bnsum_slabs(int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil,
                                   long ws_bytes) {
  if (B <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || dil != 1 || Cin % 4) return 0;
  ConvArgs a{nullptr, nullptr, nullptr, nullptr, B, Ho, Wo, Cout, H, W, Cin, R, S, stride, pad, dil, 1, 1.f, 0u, 0u};
  static unsigned char dummy;
  a.ws_ = ws_bytes > 0 ? &dummy : nullptr; a.ws_bytes_ = ws_bytes;
  if (!aligned_ok(a)) return 0;
  if (bnsum_beyond_wino(a)) return use_pointwise(a) ? fs_pointwise_stats_slabs((long)B * H * W) : fs_s2bwd_stats_slabs(B, Ho, Wo);
  if ((use_tapset(a) && !use_halo(a)) || !use_wino(a)) return 0;
  return fs_wino_stats_slabs(g_conv_precision, B, H, W, Cout, Cin);      // bwd-data: source channels = Cout, destination = Cin
}

// include/fovealseg.h: fs_conv2d_bwd_data + the BatchNorm-backward column sums of the layer that produced x (dx is that layer's dz)
int fs_conv2d_bwd_data_bnsum(const float* dy, const float* w, float* dx, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R,
                             int S, int stride, int pad, int dil, void* ws, long ws_bytes, const unsigned* w_amax, const float* bn_y,
                             const unsigned char* bn_mask, const float* bn_mean, const float* bn_invstd, float* slab, const float* add_src,
                             const unsigned char* add_mask, hipStream_t stream) {
  FS_REQUIRE((bn_y != nullptr || add_src != nullptr) && (add_mask == nullptr || add_src != nullptr));
  FS_REQUIRE(bn_y == nullptr || (bn_mean && bn_invstd && slab));
  FS_REQUIRE(fs_conv2d_bwd_data_bnsum_slabs(B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws_bytes) > 0);
  const FsBnSums bn{bn_y, bn_mask, bn_mean, bn_invstd, add_src, add_mask, nullptr, nullptr, nullptr, 0};
  return conv2d_bwd_data_impl(dy, w, dx, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, ws, ws_bytes, w_amax, &bn, slab, stream);
}

// include/fovealseg.h: fs_conv2d_bwd_weight   (dw is overwritten, or added to when accumulate != 0)
// include/fovealseg.h: fs_linear_bwd_weight_bias_ok -- 1 when the fused launch below exists for this shape in the current mode
int fs_linear_bwd_weight_bias_ok(long rows, int Cin, int Cout) { return fs_linear_wgrad_eligible(g_conv_precision, rows, Cin, Cout) ? 1 : 0; }

// include/fovealseg.h: fs_linear_bwd_weight_bias -- dW[Cin][Cout] = x^T dy and dbias[Cout] = column sums of dy in one launch
int fs_linear_bwd_weight_bias(const float* x, const float* dy, float* dw, float* dbias, long rows, int Cin, int Cout, int accumulate_w,
                              int accumulate_b, void* ws, long ws_bytes, hipStream_t stream) {
  FS_REQUIRE(x && dy && dw && dbias);
  FS_REQUIRE(fs_linear_bwd_weight_bias_ok(rows, Cin, Cout) == 1);
  if (g_deterministic) {
    // ordered split-K: partial dW tiles and partial bias sums in per-split slabs, summed in index order
    const long n = (long)Cin * Cout, cap = wgrad_slab_cap(Cin, Cout, 1, 1, 1, 0, 1);
    FS_REQUIRE(ws != nullptr && ws_bytes >= fs_linear_bwd_weight_bias_ws_bytes(Cin, Cout));
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)cap * (n + 4L * Cout) * 4, stream);
    if (e != hipSuccess) return (int)e;
    FsPartHost ph{static_cast<float*>(ws), n, cap, 0, 0};
    float* bpart = static_cast<float*>(ws) + cap * n;
    const int r = fs_linear_wgrad(x, dy, dw, dbias, rows, Cin, Cout, &ph, bpart, stream);
    if (r != FS_OK) return r;
    const int r2 = fs_wgrad_reduce(ph.base, ph.used, n, dw, accumulate_w, stream);
    if (r2 != FS_OK) return r2;
    return fs_wgrad_reduce(bpart, 4 * ph.used, Cout, dbias, accumulate_b, stream);
  }
  if (!accumulate_w) {
    const hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)Cin * Cout, stream);
    if (e != hipSuccess) return (int)e;
  }
  if (!accumulate_b) {
    const hipError_t e = hipMemsetAsync(dbias, 0, sizeof(float) * (size_t)Cout, stream);
    if (e != hipSuccess) return (int)e;
  }
  return fs_linear_wgrad(x, dy, dw, dbias, rows, Cin, Cout, nullptr, nullptr, stream);
}

static int conv2d_bwd_weight_impl(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo,
                                  int Cout, int R, int S, int stride, int pad, int dil, FsPartHost* ph, hipStream_t stream);

int fs_conv2d_bwd_weight(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo,
                         int Cout, int R, int S, int stride, int pad, int dil, int accumulate, void* ws, long ws_bytes, hipStream_t stream) {
  FS_REQUIRE(x && dy && dw && B > 0 && H > 0 && W > 0 && Cin > 0 && Cout > 0 && R > 0 && S > 0 && stride > 0);
  FS_REQUIRE(dil >= 1 && Ho == (H + 2 * pad - dil * (R - 1) - 1) / stride + 1 && Wo == (W + 2 * pad - dil * (S - 1) - 1) / stride + 1);
  if (g_deterministic) {
    // ordered split-K: every split's partial tile goes to its own slab of ws (zeroed: a split without pixels writes nothing), the slabs
    // are summed in index order -- and dw itself is only touched by that sum, so no memset of it
    const long n = (long)R * S * Cin * Cout, cap = wgrad_slab_cap(Cin, Cout, R, S, stride, pad, dil);
    FS_REQUIRE(ws != nullptr && ws_bytes >= cap * n * 4);
    hipError_t e = hipMemsetAsync(ws, 0, (size_t)cap * n * 4, stream);
    if (e != hipSuccess) return (int)e;
    FsPartHost ph{static_cast<float*>(ws), n, cap, 0, 0};
    const int r = conv2d_bwd_weight_impl(x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, &ph, stream);
    if (r != FS_OK) return r;
    return fs_wgrad_reduce(ph.base, ph.used, n, dw, accumulate, stream);
  }
  if (ws != nullptr && wgrad_store_route(Cin, Cout, R, S, stride, pad, dil) &&
      (size_t)B * H * W * Cin * 4 < 4294967000UL && (size_t)B * Ho * Wo * Cout * 4 < 4294967000UL) {
    const long n = (long)R * S * Cin * Cout, cap = wgrad_slab_cap(Cin, Cout, R, S, stride, pad, dil);
    if (ws_bytes >= cap * n * 4) {
      FsPartHost ph{static_cast<float*>(ws), n, cap, 0, 1};
      const int r = fs_wgrad_split(g_conv_precision, x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, &ph, stream);
      if (r != FS_OK) return r;
      return fs_wgrad_reduce(ph.base, ph.used, n, dw, accumulate, stream);
    }
  }
  if (!accumulate) {         // every kernel below adds its split-K partials atomically: dw = 0 first, unless the caller accumulates
    hipError_t e = hipMemsetAsync(dw, 0, sizeof(float) * (size_t)R * S * Cin * Cout, stream);
    if (e != hipSuccess) return (int)e;
  }
  return conv2d_bwd_weight_impl(x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, nullptr, stream);
}

static int part_of(FsPartHost* ph, long nslab, FsPart& out) {
  out = FsPart{nullptr, 0};
  if (ph == nullptr || ph->base == nullptr) return FS_OK;
  if (nslab > ph->cap) return FS_ERR_ARG;
  if (nslab > ph->used) ph->used = (int)nslab;
  out = FsPart{ph->base, ph->stride};
  return FS_OK;
}

static int conv2d_bwd_weight_impl(const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo,
                                  int Cout, int R, int S, int stride, int pad, int dil, FsPartHost* ph, hipStream_t stream) {
  const long P = (long)B * Ho * Wo;
  if (g_conv_precision >= 1 && fs_wgrad_split_eligible(Cin, Cout, R, S, stride, pad, dil) &&
      (size_t)B * H * W * Cin * 4 < 4294967000UL && (size_t)P * Cout * 4 < 4294967000UL)
    return fs_wgrad_split(g_conv_precision, x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, ph, stream);
  const int tiles = cdiv(Cin, 64) * cdiv(Cout, 64);
  // one filter row per workgroup for narrow layers (more workgroups, fewer atomics each), the whole 3x3
  // filter per workgroup once there are >= 9 channel tiles (measured: 64^2/128^2 87-90 TF with 3,
  // 192^2 106 TF / 960x240 87 TF / 512^2 87 TF with 9; per-tap kernel 71-83 TF).
  const int wg_mode = tiles >= 9 ? 9 : 3;
  if (R == 3 && S == 3 && (Cin % 4 == 0) && (Cout % 4 == 0) &&
      (size_t)B * H * W * Cin * 4 < 4294967000UL && (size_t)P * Cout * 4 < 4294967000UL) {
    const int ng = 9 / wg_mode;
    long ns = (1024 + (long)tiles * ng - 1) / ((long)tiles * ng);
    long mx = (P + 127) / 128;
    if (ns > mx) ns = mx;
    if (ns < 1) ns = 1;
    long pp = (P + ns - 1) / ns;
    pp = ((pp + BK - 1) / BK) * BK;
    ns = (P + pp - 1) / pp;
    WgradArgs a{x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, (int)pp, tiles, 9, (int)ns, FsPart{nullptr, 0}};
    if (part_of(ph, ns, a.part) != FS_OK) return FS_ERR_ARG;
    dim3 grid((unsigned)(tiles * ng * ns));
    if (wg_mode == 3) hipLaunchKernelGGL(conv_wgrad_taps_kernel<3>, grid, dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(conv_wgrad_taps_kernel<9>, grid, dim3(256), 0, stream, a);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  long nsplit = (2048 + (long)tiles * R * S - 1) / ((long)tiles * R * S);
  long maxsplit = (P + 255) / 256;
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit < 1) nsplit = 1;
  long pps = (P + nsplit - 1) / nsplit;
  pps = ((pps + BK - 1) / BK) * BK;
  nsplit = (P + pps - 1) / pps;
  WgradArgs a{x, dy, dw, B, H, W, Cin, Ho, Wo, Cout, R, S, stride, pad, dil, (int)pps, tiles, R * S, (int)nsplit, FsPart{nullptr, 0}};
  if (part_of(ph, nsplit, a.part) != FS_OK) return FS_ERR_ARG;
  dim3 grid((unsigned)(tiles * R * S * nsplit));
  if ((Cin % 4 == 0) && (Cout % 4 == 0))
    hipLaunchKernelGGL(conv_wgrad_kernel<true>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(conv_wgrad_kernel<false>, grid, dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // extern "C"
