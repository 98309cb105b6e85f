// 3x3 / stride-1 / pad-1 convolution (forward and bwd-data) as a halo-tiled implicit GEMM in split precision
// (conv_split.h: f16x2 = two scaled fp16 planes / 3 MFMAs per product, bf16x3 = three bf16 planes / 6 MFMAs; fp32 accumulate).
//
// The workgroup's 128 output pixels are a Ph x Pw patch; the (Ph+2) x (Pw+2) input halo of a 32-channel chunk is loaded,
// (scaled,) split and written to LDS ONCE and all 9 taps read it at shifted slot offsets (18 k16-steps between barriers).
// The weights are split ahead of time by a pack kernel into the exact order the MFMA B fragments are consumed, so B fragments
// are 16-byte global loads straight into registers (contiguous 1 KB per wave load) and never touch LDS or the VALU.  The plain
// kernel (conv_igemm_split_kernel) re-loaded and re-split its A tile for every tap: 285 VALU instructions per 24 MFMAs.
//
//   workgroup = 256 threads = 4 waves as 2 (pixel halves) x 2 (channel halves); wave tile 64 x 32*NW
//   LDS: NPL planes x 224 halo slots x 80 B (32 k + 16 B pad -> conflict-free ds_read_b128)
//
// f16x2 range handling: fp16 has 5 exponent bits, so the scale is chosen where the data is seen.
//   * activations: per (workgroup, 32-channel chunk) -- the workgroup takes the max |x| of the halo tile it has just loaded
//     (register max -> wave max -> LDS atomic max, read back behind the barrier that already separates the MFMA phase from
//     the LDS refill) and keeps a running exponent E = max over chunks; operands are scaled by 2^(14-E) (largest element in
//     [2^14, 2^15), never overflows), and when a new chunk raises E the fp32 accumulators are rescaled by the exact power of
//     two.  A chunk much smaller than an earlier one is therefore resolved relative to the accumulated magnitude -- which is
//     what an fp32 chain does as well.
//   * weights: one exponent per weight tensor (kept by the caller, fs_weight_amax_segments, or from an atomic-max pre-pass
//     into the pack header); the pack kernel writes the planes already scaled.
// The final result is acc * 2^(E-14) * 2^(Ew-14), applied in the epilogue.
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS slot (80 bytes)
constexpr int NSMAX = 224;
constexpr int NITEM = 7;
constexpr int PLANE = NSMAX * XLD;

struct HaloArgs {
  const float* src; const unsigned char* ws; const unsigned* ew; const float* bias; float* dst; float* stats;
  int B, H, W, Cs, Cd, Npad, nchunk;
  int Ph, Pw, tiles_y, tiles_x;
  int nx, ny;
  unsigned src_bytes, ws_bytes, dst_bytes;
  unsigned magic_pw, magic_wh;      // 2^32 / Pw + 1, 2^32 / (Pw + 2) + 1
  // stacked = 1: the batch is tiled as ONE image of B*(H+1) rows, a zero row after every image (the vertical padding the two
  // neighbours share), so patches need not divide H: 20x20 and 10x10 layers fill 87 % of their tile rows instead of 78 %
  int stacked, Hv;                  // Hv = H + 1
  unsigned magic_hv;
  float drop_scale; uint32_t drop_thresh, drop_key;
};

// max |w| over the weight tensor as float bits (non-negative floats order like unsigned ints)
__global__ __launch_bounds__(256) void conv_f16_amax_kernel(const float* __restrict__ w, long n, unsigned* __restrict__ out) {
  float m = 0.f;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0) atomicMax(out, __builtin_bit_cast(unsigned, m));
}

// Weight pack: Wp[g = (chunk*9 + tap)*2 + s][plane][n][j] = plane-th term of Wt[tap][k = 32*chunk + 16*s + j][n] (scaled by
// 2^(14-Ew) in f16x2), behind a HDR-byte header.
//   forward : Wt[tap][k][n] = W[tap][k][n]                  (K = Cin,  N = Cout)
//   bwd-data: Wt[tap][k][n] = W[8 - tap][n][k]              (K = Cout, N = Cin; taps flipped)
// zero for k >= K or n >= N.  One thread per (g, n) row of 16 k.
template <class P>
__global__ __launch_bounds__(256) void conv_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew,
                                                        int Cin, int Cout, int transposed, int Ks, int Ns, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int s = g & 1, tap = (g >> 1) % 9, chunk = g / 18;
  const int k0 = chunk * 32 + s * 16;
  typename P::x8 p[P::NPL][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Ns && k < Ks) v = transposed ? w[((long)(8 - tap) * Cin + n) * Cout + k] : w[((long)tap * Cin + k) * Cout + n];
    typename P::T t[P::NPL];
    P::split(v * sc, t);
#pragma unroll
    for (int pl = 0; pl < P::NPL; ++pl) p[pl][j >> 3][j & 7] = t[pl];
  }
#pragma unroll
  for (int pl = 0; pl < P::NPL; ++pl) {
    typename P::x8* o = reinterpret_cast<typename P::x8*>(wp + (((long)g * P::NPL + pl) * Npad + n) * 16);
    o[0] = p[pl][0]; o[1] = p[pl][1];
  }
}

// NW = 32-column sub-tiles per wave: the workgroup covers 64*NW output channels, so the halo image is loaded, scaled and split
// once for twice the MFMAs when NW = 2 (layers with >= 128 output channels), and an A fragment feeds 2 MFMA groups.
template <class P, int NW>
__global__ __launch_bounds__(256, (NW == 1 && P::NPL == 2) ? 3 : 2) void conv3x3_halo_kernel(HaloArgs a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  __shared__ __attribute__((aligned(16))) typename P::T Ah[NPL * PLANE];
  __shared__ __attribute__((aligned(16))) int rowpix[128];
  __shared__ unsigned amax_cell[2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;
  const int mt = wg / a.ny;
  const int n0 = (wg - mt * a.ny) * 64 * NW;
  // plain: tile (b, ty, tx) of image b.  stacked: tile (ty, tx) of the virtual image, y0 is a virtual row.
  const int tpi = a.tiles_y * a.tiles_x;
  const int b0 = a.stacked ? 0 : mt / tpi;
  const int trem = mt - b0 * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * a.Ph, x0 = tx * a.Pw;
  const int Wh = a.Pw + 2, nslots = (a.Ph + 2) * Wh;
  // (image, row) of virtual row vy; rows outside every image (gap rows, beyond the batch) get row = H (invalid)
  auto image_row = [&](int vy, int& bb, int& yy) {
    if (a.stacked) {
      const bool in = vy >= 0 && vy < a.B * a.Hv;
      bb = in ? div_small(vy, a.magic_hv) : 0;
      yy = in ? vy - bb * a.Hv : a.H;
    } else {
      bb = b0; yy = (vy >= 0 && vy < a.H) ? vy : a.H;
    }
  };

  if (tid < 128) {
    const int p = (tid & ~31) + row_perm(tid & 31);
    const int py = div_small(p, a.magic_pw), px = p - py * a.Pw;
    int bb, yy;
    image_row(y0 + py, bb, yy);
    const bool live = p < a.Ph * a.Pw && yy < a.H && x0 + px < a.W;
    rowpix[tid] = live ? ((bb * a.H + yy) * a.W + x0 + px) : -1;
  }
  if (tid < 2) amax_cell[tid] = 0u;
  const int q = tid & 7;
  int goff[NITEM];
  {
    const int q32 = div_small(32, a.magic_wh), r32 = 32 - q32 * Wh;
    int hy = div_small(tid >> 3, a.magic_wh), hx = (tid >> 3) - hy * Wh;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int ix = x0 + hx - 1;
      int bb, iy;
      image_row(y0 + hy - 1, bb, iy);
      const bool ok = (tid >> 3) + 32 * i < nslots && iy < a.H && ix >= 0 && ix < a.W;
      goff[i] = ok ? ((bb * a.H + iy) * a.W + ix) * a.Cs + 4 * q : -1;
      hx += r32; hy += q32;
      if (hx >= Wh) { hx -= Wh; ++hy; }
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);

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
  auto tile_amax = [&](int cell) {       // max |x| of the loaded halo registers -> LDS cell
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NITEM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ra[i][e]));
    m = wave_max(m);
    if (lane == 0) atomicMax(&amax_cell[cell], __builtin_bit_cast(unsigned, m));
  };
  auto store_halo = [&](float sc) {
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (tid >> 3) + 32 * i;
      X4 p[NPL];
      P::split4(P::SCALED ? ra[i] * sc : ra[i], p);
      const int o = slot * XLD + 4 * q;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ah[pl * PLANE + o]) = p[pl];
    }
  };

  int rowbase[2][3];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int p = 64 * wm + 32 * mi + row_perm(l31);
    const bool live = p < a.Ph * a.Pw;
    const int py = live ? div_small(p, a.magic_pw) : 0, px = live ? p - py * a.Pw : 0;
#pragma unroll
    for (int r = 0; r < 3; ++r) rowbase[mi][r] = ((py + r) * Wh + px) * XLD + 8 * lh;
  }
  const int bvoff = HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;      // sub-tile j: + j * 64 columns = j * 2048 bytes
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = a.nchunk * 18;

  X8 fa[2][2][NPL];       // [buffer][mi][plane]
  X8 fb[3][NW][NPL];      // [ring slot][sub-tile][plane]: fragments run 2 steps ahead of the MFMAs
  auto load_b = [&](int g, X8 (&dst)[NW][NPL]) {
    const int gg = g < G ? g : G - 1;            // wave-uniform: the stream position goes in the scalar offset operand
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, bvoff + j * 2048, gg * step_bytes + pl * plane_bytes, 0);
        dst[j][pl] = __builtin_bit_cast(X8, v);
      }
  };

  f32x16 acc[2][NW];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi)
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][j][r] = 0.f;
  int E = EMIN;
  load_b(0, fb[0]);
  load_b(1, fb[1]);
  load_halo(0);
  __syncthreads();                        // amax cells zeroed before the first atomic
  int g = 0;
  for (int chunk = 0; chunk < a.nchunk; ++chunk) {
    if (P::SCALED) tile_amax(chunk & 1);
    __syncthreads();                      // amax complete; every wave has finished reading the previous image
    if (P::SCALED) {
      const int ec = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[chunk & 1]));
      if (ec > E) {
        if (chunk > 0) {                  // accumulators (still zero in the first chunk) move to the new unit 2^(ec-14)
          const float f = pow2f(E - ec);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][j][r] *= f;
        }
        E = ec;
      }
      if (tid == 0) amax_cell[(chunk + 1) & 1] = 0u;
    }
    store_halo(pow2f(14 - E));
    __syncthreads();
    if (chunk + 1 < a.nchunk) load_halo(chunk + 1);
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) fa[0][mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase[mi][0]]);
#pragma unroll
    for (int step = 0; step < 18; ++step) {
      if (step + 1 < 18) {
        const int tap = (step + 1) >> 1, s2 = (step + 1) & 1, r = tap / 3, s = tap - 3 * r;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            fa[(step + 1) & 1][mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase[mi][r] + s * XLD + 16 * s2]);
      }
      load_b(g + 2, fb[(step + 2) % 3]);
      __builtin_amdgcn_sched_barrier(0);
      const X8(&A)[2][NPL] = fa[step & 1];
      const X8(&Bf)[NW][NPL] = fb[step % 3];
#pragma unroll
      for (int j = 0; j < NW; ++j)
#pragma unroll
        for (int t = 0; t < P::NTERM; ++t) {       // smallest terms first, the two pixel halves interleaved
          acc[0][j] = P::mfma(A[0][P::ta(t)], Bf[j][P::tb(t)], acc[0][j]);
          acc[1][j] = P::mfma(A[1][P::ta(t)], Bf[j][P::tb(t)], acc[1][j]);
        }
      __builtin_amdgcn_sched_barrier(0);
      ++g;
    }
  }

  // ---- epilogue ----
  __syncthreads();
  // f16x2: acc * 2^(E-14) * 2^(Ew-14), one factor when the combined exponent is a normal float, two otherwise
  float f1 = 1.f, f2 = 1.f;
  if (P::SCALED) {
    const int Ew = exponent_of_bits(*a.ew);
    const int es = E + Ew - 28;
    const bool one = es >= -126 && es <= 127;
    f1 = one ? pow2f(es) : pow2f(E - 14);
    f2 = one ? 1.f : pow2f(Ew - 14);
  }
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
  float csum[NW], csq[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    csum[j] = 0.f; csq[j] = 0.f;
    const int n = n0 + 64 * j + 32 * wn + l31;
    if (n >= a.Cd) continue;
    const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const i32x4 pix = *reinterpret_cast<const i32x4*>(&rowpix[64 * wm + 32 * mi + 8 * rg + 4 * lh]);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int r = 4 * rg + ri;
          const bool live = pix[ri] >= 0;
          const unsigned e = (unsigned)pix[ri] * (unsigned)a.Cd + (unsigned)n;      // element index (< 2^30: dst_bytes < 4 GB)
          float v = P::SCALED ? fmaf(acc[mi][j][r] * f2, f1, bv) : acc[mi][j][r] + bv;
          if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
          v = live ? v : 0.f;
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
          csum[j] += v; csq[j] += v * v;
        }
      }
    }
  }
  if (a.stats != nullptr) {
    float* red = reinterpret_cast<float*>(&Ah[0]);     // [wm][64*NW cols][2]; the halo image is dead since the barrier above
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const float s1 = csum[j] + __shfl_xor(csum[j], 32, 64), s2 = csq[j] + __shfl_xor(csq[j], 32, 64);
      const int col = 64 * j + 32 * wn + l31;
      if (lh == 0) { red[(wm * 64 * NW + col) * 2] = s1; red[(wm * 64 * NW + col) * 2 + 1] = s2; }
    }
    __syncthreads();
    for (int t = tid; t < 128 * NW; t += 256) {
      const int col = t >> 1, which = t & 1;
      const float v = red[col * 2 + which] + red[(64 * NW + col) * 2 + which];
      if (n0 + col < a.Cd) a.stats[((long)mt * a.Cd + n0 + col) * 2 + which] = v;
    }
  }
}

}  // namespace

thread_local int fs_ws_mode_tls = 0;

const unsigned* fs_f16_weight_amax(const float* w, long n, void* ws, const unsigned* w_amax, hipStream_t stream, int* err) {
  *err = FS_OK;
  if (w_amax != nullptr) return w_amax;          // the caller keeps max|w| of this tensor up to date (fs_weight_amax_segments)
  if (fs_ws_mode_tls == FS_WS_RUN_ONLY) return reinterpret_cast<const unsigned*>(ws);      // left there by the pack-only call
  hipError_t e = hipMemsetAsync(ws, 0, 4, stream);
  if (e != hipSuccess) { *err = (int)e; return nullptr; }
  int ab = cdiv(n, 256 * 8); if (ab > 256) ab = 256;
  hipLaunchKernelGGL(conv_f16_amax_kernel, dim3(ab), dim3(256), 0, stream, w, n, reinterpret_cast<unsigned*>(ws));
  if (hipGetLastError() != hipSuccess) { *err = FS_ERR_ARG; return nullptr; }
  return reinterpret_cast<const unsigned*>(ws);
}

// max|w| bits of every parameter of a flat arena in one launch: block (p, j) strides over parameter p
__global__ __launch_bounds__(256) void weight_amax_segments_kernel(const float* __restrict__ arena, const long* __restrict__ offsets,
                                                                   const long* __restrict__ sizes, unsigned* __restrict__ out) {
  const int p = blockIdx.x;
  const float* w = arena + offsets[p];
  const long n = sizes[p];
  const long n4 = (((size_t)w & 15) == 0) ? n / 4 : 0;  // 16-byte loads, two in flight per trip
  if ((long)blockIdx.y * (n4 ? 1024 : 256) >= n) return;   // small parameters need one block (block 0 also takes the tail)
  float m = 0.f;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(w);
  const long stride = (long)gridDim.y * 256;
  for (long i = (long)blockIdx.y * 256 + threadIdx.x; i < n4; i += 2 * stride) {
    const long i2 = i + stride < n4 ? i + stride : i;
    const f32x4 a = w4[i], b = w4[i2];
#pragma unroll
    for (int e = 0; e < 4; ++e) m = fmaxf(m, fmaxf(fabsf(a[e]), fabsf(b[e])));
  }
  for (long i = 4 * n4 + (long)blockIdx.y * 256 + threadIdx.x; i < n; i += stride) m = fmaxf(m, fabsf(w[i]));
  m = wave_max(m);
  if ((threadIdx.x & 63) == 0 && m > 0.f) atomicMax(&out[p], __builtin_bit_cast(unsigned, m));
}

int fs_weight_amax_segments_impl(const float* arena, const long* offsets, const long* sizes, int nparams, unsigned* out, hipStream_t stream) {
  hipError_t e = hipMemsetAsync(out, 0, sizeof(unsigned) * (size_t)nparams, stream);
  if (e != hipSuccess) return (int)e;
  hipLaunchKernelGGL(weight_amax_segments_kernel, dim3(nparams, 32), dim3(256), 0, stream, arena, offsets, sizes, out);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

static inline int halo_nw(int Cd) { return Cd >= 128 ? 2 : 1; }      // 32-column sub-tiles per wave

// Ph x Pw <= 128 output pixels, halo (Ph+2)(Pw+2) <= NSMAX; fewest tiles per image, then smallest halo.
static void choose_patch(int H, int W, int& Ph, int& Pw) {
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

// Tiling of the output pixels: per image or over the stacked batch, whichever needs fewer 128-row tiles.
static void halo_plan(int B, int H, int W, int& Ph, int& Pw, int& stacked, int& tiles_y, int& tiles_x, int& nx) {
  choose_patch(H, W, Ph, Pw);
  stacked = 0;
  tiles_y = cdiv(H, Ph); tiles_x = cdiv(W, Pw);
  nx = B * tiles_y * tiles_x;
  const long rows = (long)B * (H + 1);
  if (rows >= 65536) return;                       // div_small range
  for (int pw = 4; pw <= 64 && pw <= W + 3; ++pw) {
    int ph = 128 / pw;
    while (ph > 1 && (ph + 2) * (pw + 2) > NSMAX) --ph;
    if (ph < 1 || (ph + 2) * (pw + 2) > NSMAX) continue;
    const long t = (long)cdiv(rows, ph) * cdiv(W, pw);
    if (t < nx) { nx = (int)t; stacked = 1; Ph = ph; Pw = pw; tiles_y = cdiv(rows, ph); tiles_x = cdiv(W, pw); }
  }
}

template <class P>
static int run_halo(HaloArgs& a, const float* w, void* ws, const unsigned* w_amax, int Cin, int Cout, int transposed, int nw,
                    hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(w, (long)9 * Cin * Cout, ws, w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.nchunk * 18 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((conv_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws),
                       a.ew, Cin, Cout, transposed, a.Cs, a.Cd, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
  if (nw == 2) hipLaunchKernelGGL((conv3x3_halo_kernel<P, 2>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv3x3_halo_kernel<P, 1>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

bool fs_halo_eligible(int H, int W, int Cs, int Cd, int R, int S, int stride, int pad, int dil) {
  (void)H; (void)W;
  return R == 3 && S == 3 && stride == 1 && pad == 1 && dil == 1 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 32;
}

int fs_halo_stats_slabs(int B, int H, int W) {
  int Ph, Pw, st, ty, tx, nx;
  halo_plan(B, H, W, Ph, Pw, st, ty, tx, nx);
  return nx;
}

long fs_halo_pack_bytes(int mode, int Cs, int Cd) {
  const int nw = halo_nw(Cd), npl = mode == 2 ? 2 : 3;
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 64 * nw - 1) / (64 * nw)) * 64 * nw;
  return HDR + nchunk * 18 * npl * Npad * 16 * 2;
}

int fs_halo_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                    int B, int H, int W, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh,
                    uint32_t drop_key, hipStream_t stream) {
  HaloArgs a;
  a.src = src; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = dst; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cs = Cs; a.Cd = Cd;
  const int nwp = halo_nw(Cd);
  a.Npad = ((Cd + 64 * nwp - 1) / (64 * nwp)) * 64 * nwp;      // row count of the pack (fs_halo_pack_bytes)
  a.nchunk = (Cs + 31) / 32;
  halo_plan(B, H, W, a.Ph, a.Pw, a.stacked, a.tiles_y, a.tiles_x, a.nx);
  a.Hv = H + 1;
  a.magic_hv = div_magic(a.Hv);
  // 128-column workgroups (two sub-tiles per wave) while there are still ~1.75 of them per CU; measured on 512->512 @ 10x10
  // (256 such workgroups): 144 us against 130 us with 512 workgroups of 64 columns; 448 on 256->256 @ 20x20: +0.7 % on the step
  const int nw = (nwp == 2 && (long)a.nx * (a.Npad / 128) >= 440) ? 2 : 1;
  a.ny = nw == 2 ? a.Npad / 128 : (Cd + 63) / 64;
  a.magic_pw = div_magic(a.Pw);
  a.magic_wh = div_magic(a.Pw + 2);
  a.src_bytes = (unsigned)((size_t)B * H * W * Cs * 4);
  a.dst_bytes = (unsigned)((size_t)B * H * W * Cd * 4);
  const long pack_bytes = fs_halo_pack_bytes(mode, Cs, Cd);
  if (pack_bytes >= 2147483647L || (size_t)B * H * W * Cs * 4 >= 4294967000UL || (size_t)B * H * W * Cd * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.ws_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  return mode == 2 ? run_halo<PrecF16>(a, w, ws, w_amax, Cin, Cout, transposed, nw, stream)
                   : run_halo<PrecX3>(a, w, ws, w_amax, Cin, Cout, transposed, nw, stream);
}
