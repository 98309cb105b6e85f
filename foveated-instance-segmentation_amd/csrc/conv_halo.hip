// 3x3 / stride-1 / pad-1 convolution (forward and bwd-data) as a halo-tiled implicit GEMM on the bf16 MFMA
// pipe in split precision ("bf16x3": x = x1 + x2 + x3 with three bf16 terms, six v_mfma_f32_32x32x16_bf16
// per fp32 product, fp32 accumulation -- same arithmetic as conv_igemm_x3_kernel in conv.hip).
//
// Why a second kernel: in conv_igemm_x3_kernel every workgroup re-loads and re-splits its 128 x 32 A tile
// for each of the 9 filter taps, and the split (about 6 VALU instructions per element) costs more issue
// slots than the MFMAs it feeds (SQ counters: 285 VALU instructions per 24 MFMAs, MFMA pipe 30 % busy).
// Here the workgroup's 128 output pixels are a Ph x Pw patch of one image; the (Ph+2) x (Pw+2) input halo
// of a 32-channel chunk is loaded, split and written to LDS ONCE and all 9 taps read it at shifted slot
// offsets (216 MFMAs per wave between barriers).  The weights are split ahead of time by a small pack
// kernel into the exact order the MFMA B fragments are consumed, so B fragments are plain 16-byte global
// loads straight into registers (contiguous 1 KB per wave load) and never touch LDS or the VALU.
//
//   workgroup = 256 threads = 4 waves as 2 (pixel halves) x 2 (channel halves); wave tile 64 x 32
//   LDS: 3 bf16 planes x 224 halo slots x 80 B (32 k + 16 B pad -> conflict-free ds_read_b128) = 53 760 B
#include "common.h"
#include "conv_halo.h"

namespace {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr int XLD = 40;            // bf16 per LDS slot (80 bytes)
constexpr int NSMAX = 224;         // halo slots per plane (7 x 32: every thread stores exactly NITEM quads)
constexpr int NITEM = 7;           // NSMAX * 8 quads / 256 threads
constexpr int PLANE = NSMAX * XLD; // bf16 per plane
constexpr unsigned OOB = 0xFFFFFFF0u;

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}

__device__ __forceinline__ void split3(float x, __bf16& a, __bf16& b, __bf16& c) {
  a = (__bf16)x;
  const float r = x - (float)a;
  b = (__bf16)r;
  c = (__bf16)(r - (float)b);
}

// ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (per 32-lane half).  Map the
// 32 rows of an MFMA tile to patch pixels so that each group reads 16 CONSECUTIVE pixels (conflict-free 80-B rows).
__device__ __forceinline__ int row_perm(int l) {
  const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28;
  if (!g1) return l < 4 ? l : (l < 16 ? l - 8 : l - 12);
  return 16 + (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16));
}

struct HaloArgs {
  const float* src;     // (B,H,W,Cs) fp32 NHWC
  const __bf16* wp;     // packed weights [nchunk*18][3][Npad][16]
  const float* bias;    // may be null
  float* dst;           // (B,H,W,Cd)
  float* stats;         // optional [nx][Cd][2]
  int B, H, W, Cs, Cd, Npad, nchunk;
  int Ph, Pw, tiles_y, tiles_x;
  int nx, ny;
  unsigned src_bytes, wp_bytes;
  float drop_scale; uint32_t drop_thresh, drop_key;
};

// Weight pack: Wp[g = (chunk*9 + tap)*2 + s][plane][n][j] = plane-th bf16 term of Wt[tap][k = 32*chunk + 16*s + j][n]
//   forward : Wt[tap][k][n] = W[tap][k][n]                  (K = Cin,  N = Cout)
//   bwd-data: Wt[tap][k][n] = W[8 - tap][n][k]              (K = Cout, N = Cin; taps flipped)
// zero for k >= K or n >= N.  One thread per (g, n) row of 16 k.
__global__ __launch_bounds__(256) void conv_pack_x3_kernel(const float* __restrict__ w, __bf16* __restrict__ wp, int Cin, int Cout,
                                                           int transposed, int Ks, int Ns, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int s = g & 1, tap = (g >> 1) % 9, chunk = g / 18;
  const int k0 = chunk * 32 + s * 16;
  bf16x8 p[3][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Ns && k < Ks) v = transposed ? w[((long)(8 - tap) * Cin + n) * Cout + k] : w[((long)tap * Cin + k) * Cout + n];
    __bf16 a, b, c;
    split3(v, a, b, c);
    p[0][j >> 3][j & 7] = a; p[1][j >> 3][j & 7] = b; p[2][j >> 3][j & 7] = c;
  }
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) {
    bf16x8* o = reinterpret_cast<bf16x8*>(wp + (((long)g * 3 + pl) * Npad + n) * 16);
    o[0] = p[pl][0]; o[1] = p[pl][1];
  }
}

__global__ __launch_bounds__(256) void conv3x3_halo_x3_kernel(HaloArgs a) {
  __shared__ __attribute__((aligned(16))) __bf16 Ah[3 * PLANE];
  __shared__ __attribute__((aligned(16))) int rowpix[128];      // output pixel of every tile row (MFMA row order), -1 = dead

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // XCD-aware tile order (see conv_igemm_affine_kernel): each XCD gets a contiguous range of tiles.
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
  const int mt = wg / a.ny;
  const int n0 = (wg - mt * a.ny) * 64;
  const int tpi = a.tiles_y * a.tiles_x;
  const int b = mt / tpi;
  const int trem = mt - b * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * a.Ph, x0 = tx * a.Pw;
  const int Wh = a.Pw + 2, nslots = (a.Ph + 2) * Wh;

  if (tid < 128) {
    const int p = (tid & ~31) + row_perm(tid & 31);
    const int py = p / a.Pw, px = p - py * a.Pw;
    const bool live = p < a.Ph * a.Pw && y0 + py < a.H && x0 + px < a.W;
    rowpix[tid] = live ? ((b * a.H + y0 + py) * a.W + x0 + px) : -1;
  }
  // ---- halo loader: item = tid + 256*i -> (slot = item>>3, channel quad q = item&7); slots advance by 32 per item ----
  const int q = tid & 7;
  int goff[NITEM];
  {
    const int q32 = 32 / Wh, r32 = 32 - q32 * Wh;
    int hy = (tid >> 3) / Wh, hx = (tid >> 3) - hy * Wh;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int iy = y0 + hy - 1, ix = x0 + hx - 1;
      const bool ok = (tid >> 3) + 32 * i < nslots && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
      goff[i] = ok ? ((b * a.H + iy) * a.W + ix) * a.Cs + 4 * q : -1;
      hx += r32; hy += q32;
      if (hx >= Wh) { hx -= Wh; ++hy; }
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.wp, a.wp_bytes);

  f32x4 ra[NITEM];
  auto load_halo = [&](int chunk) {
    const int c0 = chunk * 32;
    const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const bool ok = cok && goff[i] >= 0;
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + c0) * 4u) : (int)OOB, 0, 0);
      ra[i] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto store_halo = [&]() {
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (tid >> 3) + 32 * i;
      bf16x4 p0, p1, p2;
#pragma unroll
      for (int e = 0; e < 4; ++e) { __bf16 x, y, z; split3(ra[i][e], x, y, z); p0[e] = x; p1[e] = y; p2[e] = z; }
      const int o = slot * XLD + 4 * q;
      *reinterpret_cast<bf16x4*>(&Ah[o]) = p0;
      *reinterpret_cast<bf16x4*>(&Ah[PLANE + o]) = p1;
      *reinterpret_cast<bf16x4*>(&Ah[2 * PLANE + o]) = p2;
    }
  };

  // ---- A fragment rows: tile row 64*wm + 32*mi + l31 -> patch pixel p = 64*wm + 32*mi + row_perm(l31); top-left halo slot = (py, px) ----
  int rowbase[2][3];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int p = 64 * wm + 32 * mi + row_perm(l31);
    const bool live = p < a.Ph * a.Pw;
    const int py = live ? p / a.Pw : 0, px = live ? p - py * a.Pw : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) rowbase[mi][r] = ((py + r) * Wh + px) * XLD + 8 * lh;
  }
  // ---- B fragments: lane (l31 -> n, lh -> k half) reads 16 B at Wp[g][plane][n0 + 32*wn + l31][8*lh] ----
  const int bvoff = ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;       // bytes
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = 3 * plane_bytes;
  const int G = a.nchunk * 18;

  bf16x8 fa[2][2][3];   // [buffer][mi][plane]
  bf16x8 fb[3][3];      // [buffer][plane]
  auto load_b = [&](int g, bf16x8 (&dst)[3]) {
    const int gg = g < G ? g : G - 1;
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, bvoff + gg * step_bytes + pl * plane_bytes, 0, 0);
      dst[pl] = __builtin_bit_cast(bf16x8, v);
    }
  };

  f32x16 acc0 = {0}, acc1 = {0};
  load_b(0, fb[0]);
  load_b(1, fb[1]);
  load_halo(0);
  int g = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    __syncthreads();
    store_halo();
    __syncthreads();
    if (chunk + 1 < a.nchunk) load_halo(chunk + 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) fa[0][mi][pl] = *reinterpret_cast<const bf16x8*>(&Ah[pl * PLANE + rowbase[mi][0]]);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
      if (step + 1 < 18) {
        const int tap = (step + 1) >> 1, s2 = (step + 1) & 1, r = tap / 3, s = tap - 3 * r;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int pl = 0; pl < 3; ++pl)
            fa[(step + 1) & 1][mi][pl] = *reinterpret_cast<const bf16x8*>(&Ah[pl * PLANE + rowbase[mi][r] + s * XLD + 16 * s2]);
      }
      load_b(g + 2, fb[(step + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      const bf16x8(&A)[2][3] = fa[step & 1];
      const bf16x8(&Bf)[3] = fb[step % 3];
      // smallest cross terms first
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], Bf[2], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], Bf[2], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][1], Bf[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][1], Bf[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][2], Bf[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][2], Bf[0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], Bf[1], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], Bf[1], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][1], Bf[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][1], Bf[0], acc1, 0, 0, 0);
      acc0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[0][0], Bf[0], acc0, 0, 0, 0);
      acc1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[1][0], Bf[0], acc1, 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
      ++g;
    }
  }

  // ---- epilogue: bias, dropout, store, optional BatchNorm partial sums ----
  __syncthreads();      // rowpix visible; every wave is done with the halo image
  const int n = n0 + 32 * wn + l31;
  float csum = 0.f, csq = 0.f;
  if (n < a.Cd) {
    const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const i32x4 pix = *reinterpret_cast<const i32x4*>(&rowpix[64 * wm + 32 * mi + 8 * rg + 4 * lh]);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          if (pix[ri] < 0) continue;
          const int r = 4 * rg + ri;
          float v = (mi == 0 ? acc0[r] : acc1[r]) + bv;
          const long e = (long)pix[ri] * a.Cd + n;
          if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
          a.dst[e] = v;
          csum += v; csq += v * v;
        }
      }
    }
  }
  if (a.stats != nullptr) {
    float* red = reinterpret_cast<float*>(&Ah[0]);     // [wm][64 cols][2]; the halo image is dead since the barrier above
    const float s1 = csum + __shfl_xor(csum, 32, 64), s2 = csq + __shfl_xor(csq, 32, 64);
    if (lh == 0) { red[(wm * 64 + 32 * wn + l31) * 2] = s1; red[(wm * 64 + 32 * wn + l31) * 2 + 1] = s2; }
    __syncthreads();
    if (tid < 128) {
      const int col = tid >> 1, which = tid & 1;
      const float v = red[col * 2 + which] + red[(64 + col) * 2 + which];
      if (n0 + col < a.Cd) a.stats[((long)mt * a.Cd + n0 + col) * 2 + which] = v;
    }
  }
}

// Patch choice: Ph x Pw <= 128 output pixels, halo (Ph+2)(Pw+2) <= NSMAX; fewest tiles per image, then smallest halo.
void choose_patch(int H, int W, int& Ph, int& Pw) {
  long best = -1;
  Ph = 8; Pw = 16;
  for (int pw = 4; pw <= 64 && pw <= W + 3; ++pw) {
    int ph = 128 / pw;
    if (ph > H) ph = H;
    while (ph > 1 && (ph + 2) * (pw + 2) > NSMAX) --ph;
    if (ph < 1 || (ph + 2) * (pw + 2) > NSMAX) continue;
    const long tiles = (long)cdiv(H, ph) * cdiv(W, pw);
    const long cost = tiles * 1000 + (ph + 2) * (pw + 2);
    if (best < 0 || cost < best) { best = cost; Ph = ph; Pw = pw; }
  }
}

}  // namespace

void fs_halo_patch(int H, int W, int* Ph, int* Pw) { choose_patch(H, W, *Ph, *Pw); }

bool fs_halo_eligible(int H, int W, int Cs, int Cd, int R, int S, int stride, int pad, int dil) {
  (void)H; (void)W;
  return R == 3 && S == 3 && stride == 1 && pad == 1 && dil == 1 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 32;
}

long fs_halo_pack_bytes(int Cs, int Cd) {
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 63) / 64) * 64;
  return nchunk * 18 * 3 * Npad * 16 * 2;
}

int fs_halo_stats_slabs(int B, int H, int W) {
  int Ph, Pw;
  choose_patch(H, W, Ph, Pw);
  return B * cdiv(H, Ph) * cdiv(W, Pw);
}

int fs_halo_conv3x3(const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, int B, int H, int W,
                    int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh, uint32_t drop_key,
                    hipStream_t stream) {
  HaloArgs a;
  a.src = src; a.wp = reinterpret_cast<const __bf16*>(ws); a.bias = bias; a.dst = dst; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cs = Cs; a.Cd = Cd;
  a.Npad = ((Cd + 63) / 64) * 64;
  a.nchunk = (Cs + 31) / 32;
  choose_patch(H, W, a.Ph, a.Pw);
  a.tiles_y = cdiv(H, a.Ph); a.tiles_x = cdiv(W, a.Pw);
  a.nx = B * a.tiles_y * a.tiles_x;
  a.ny = a.Npad / 64;
  a.src_bytes = (unsigned)((size_t)B * H * W * Cs * 4);
  const long pack_bytes = fs_halo_pack_bytes(Cs, Cd);
  if (pack_bytes >= 2147483647L || (size_t)B * H * W * Cs * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.wp_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  const long total = (long)a.nchunk * 18 * a.Npad;
  hipLaunchKernelGGL(conv_pack_x3_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<__bf16*>(ws),
                     Cin, Cout, transposed, Cs, Cd, a.Npad, total);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(conv3x3_halo_x3_kernel, dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
