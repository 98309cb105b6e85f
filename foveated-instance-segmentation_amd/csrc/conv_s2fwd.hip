// Forward of a 3x3 / stride 2 / pad 1 convolution with ONE LDS refill per 32-channel chunk (round 4; split precision of conv_split.h).
//
// Output pixel q reads input rows 2 q + r - 1: r = 1 an EVEN row (2 q), r = 0 / 2 the ODD rows 2 q - 1 / 2 q + 1; columns alike.  The
// tap-class kernel (conv_tapset.hip) walks the four input parity planes as four classes -- four refills per chunk, two barriers and a
// split phase around 24-96 MFMAs per wave each: every strided layer sits near 100 TF whatever its K (profiles/r04/tapset_nw_ab.txt).
// Here the halo of a Ph x Pw <= 64-pixel output patch lives in LDS as its four parity planes at once
//     (odd, odd) (Ph+1) x (Pw+1)   (odd, even) (Ph+1) x Pw   (even, odd) Ph x (Pw+1)   (even, even) Ph x Pw        <= 320 slots,
// tap (r, s) reads its plane at the dense in-plane position (py + [r == 2], px + [s == 2]) -- the conflict-free 16-pixel ds_read_b128
// groups of the other halo kernels at a per-tap constant offset -- and all nine taps run on one refill: 108 * NW MFMAs per wave between
// two barriers in bf16x3 (NW = 2: 128-column workgroups on layers with >= 128 output channels).  Same epilogue as the tap-class kernel
// (bias, dropout hash, BatchNorm partial sums per workgroup).  Counterpart of conv_s2bwd.hip.
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS slot (80 bytes)
constexpr int NSMAX = 320;         // slots over the four planes
constexpr int NITEM = 10;          // NSMAX * 8 quads / 256 threads
constexpr int PLANE = NSMAX * XLD;

struct F2Args {
  const float* src; const unsigned char* ws; const unsigned* ew; const float* bias; float* dst; float* stats;
  int B, Hs, Ws, Cs, Hd, Wd, Cd;
  int Npad, nchunk;
  int Ph, Pw, tiles_y, tiles_x, nx, ny;
  int base1, base2, base3, nslots;         // first slot of planes 1..3 (plane 0 starts at 0)
  unsigned src_bytes, wp_bytes, dst_bytes;
  unsigned magic_pw, magic_pw1;            // div_small1 magics of Pw and Pw + 1
  float drop_scale; uint32_t drop_thresh, drop_key;
};

// Wp[g = 2 * (chunk * 9 + t) + s2][plane][n][j] = plane-th term of W[t / 3][t % 3][k = 32 chunk + 16 s2 + j][n]  (K = Cin, N = Cout)
template <class P>
__global__ __launch_bounds__(256) void conv_s2fwd_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew,
                                                              int Cin, int Cout, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int s2 = g & 1, T = g >> 1;
  const int chunk = T / 9, t = T - chunk * 9;
  const int k0 = chunk * 32 + s2 * 16;
  typename P::x8 p[P::NPL][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Cout && k < Cin) v = w[((long)t * Cin + k) * Cout + n];
    typename P::T tt[P::NPL];
    P::split(v * sc, tt);
#pragma unroll
    for (int pl = 0; pl < P::NPL; ++pl) p[pl][j >> 3][j & 7] = tt[pl];
  }
#pragma unroll
  for (int pl = 0; pl < P::NPL; ++pl) {
    typename P::x8* o = reinterpret_cast<typename P::x8*>(wp + (((long)g * P::NPL + pl) * Npad + n) * 16);
    o[0] = p[pl][0]; o[1] = p[pl][1];
  }
}

template <class P, int NW>
__global__ __launch_bounds__(256, 2) void conv_s2fwd_kernel(F2Args a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];        // NPL * PLANE elements of the split halo planes
  typename P::T* const Ah = reinterpret_cast<typename P::T*>(smem);
  __shared__ __attribute__((aligned(16))) int rowpix[64];
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
  const int tpi = a.tiles_y * a.tiles_x;
  const int b = mt / tpi;
  const int trem = mt - b * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * a.Ph, x0 = tx * a.Pw;
  const int npix = a.Ph * a.Pw;

  if (tid < 64) {
    const int p = (tid & ~31) + row_perm(tid & 31);
    const int py = div_small1(p, a.magic_pw), px = p - py * a.Pw;
    const bool live = p < npix && y0 + py < a.Hd && x0 + px < a.Wd;
    rowpix[tid] = live ? (b * a.Hd + y0 + py) * a.Wd + x0 + px : -1;
  }
  // A fragment row of this lane -> patch pixel -> LDS element offset inside a plane of row width Pw (rb[0]) / Pw + 1 (rb[1])
  int rb[2];
  {
    const int p = 32 * wm + row_perm(l31);
    const bool live = p < npix;
    const int py = live ? div_small1(p, a.magic_pw) : 0, px = live ? p - py * a.Pw : 0;
    rb[0] = (py * a.Pw + px) * XLD + 8 * lh;
    rb[1] = (py * (a.Pw + 1) + px) * XLD + 8 * lh;
  }
  // tap (r, s): plane = 2 * [r == 1] + [s == 1]; in-plane offset ([r == 2], [s == 2]); element offset of the tap's first slot
  auto tap_off = [&](int r, int s) -> int {
    const int pl = 2 * (r == 1) + (s == 1);
    const int base = pl == 0 ? 0 : (pl == 1 ? a.base1 : (pl == 2 ? a.base2 : a.base3));
    const int wc = a.Pw + (s != 1);
    return (base + (r == 2 ? wc : 0) + (s == 2 ? 1 : 0)) * XLD;
  };

  const int q = tid & 7;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.wp_bytes);
  if (tid < 2) amax_cell[tid] = 0u;

  // ---- halo loader: slot (tid >> 3) + 32 i over the four planes, channel quad q; one chunk ahead of the MFMA loop ----
  int goff[NITEM];
  f32x4 ra[NITEM];
#pragma unroll
  for (int i = 0; i < NITEM; ++i) {
    const int slot = (tid >> 3) + 32 * i;
    goff[i] = -1;
    if (slot < a.nslots) {
      const int pl = slot >= a.base3 ? 3 : (slot >= a.base2 ? 2 : (slot >= a.base1 ? 1 : 0));
      const int base = pl == 0 ? 0 : (pl == 1 ? a.base1 : (pl == 2 ? a.base2 : a.base3));
      const bool coleven = pl & 1, roweven = pl >> 1;
      const int local = slot - base;
      const int hy = div_small1(local, coleven ? a.magic_pw : a.magic_pw1), hx = local - hy * (a.Pw + (coleven ? 0 : 1));
      const int sy = 2 * (y0 + hy) - (roweven ? 0 : 1), sx = 2 * (x0 + hx) - (coleven ? 0 : 1);
      if (sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws) goff[i] = ((b * a.Hs + sy) * a.Ws + sx) * a.Cs + 4 * q;
    }
  }
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
  auto tile_amax = [&](int cell) {
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

  // ---- B fragments: step g = 2 * (chunk * 9 + t) + s2; sub-tile j of the wave: + 64 columns ----
  const int bvoff = HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = 18 * a.nchunk;
  auto load_b = [&](int g, X8 (&dst)[NW][NPL]) {
    const int gg = g < G ? g : G - 1;
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, bvoff + j * 2048, gg * step_bytes + pl * plane_bytes, 0);
        dst[j][pl] = __builtin_bit_cast(X8, v);
      }
  };

  f32x16 acc[NW];
#pragma unroll
  for (int j = 0; j < NW; ++j)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
  X8 fa[2][NPL];               // [k16 step][plane]
  X8 fb[2][2][NW][NPL];        // [tap parity][k16 step][sub-tile][plane]
  int E = EMIN, par = 0;

  auto read_a = [&](int off, int s2, X8 (&dst)[NPL]) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + off + 16 * s2]);
  };
  int toff[9];                 // element offset of the lane's pixel in tap t's plane (wave-uniform part + the lane's row base for that width)
#pragma unroll
  for (int t = 0; t < 9; ++t) toff[t] = tap_off(t / 3, t % 3) + ((t % 3) == 1 ? rb[0] : rb[1]);

  load_b(0, fb[0][0]);
  load_b(1, fb[0][1]);
  load_halo(0);
  __syncthreads();                        // amax cells zeroed before the first atomic; rowpix written
  for (int chunk0 = 0; chunk0 < a.nchunk; chunk0 += 2) {
#pragma unroll
    for (int cc = 0; cc < 2; ++cc) {
      const int chunk = chunk0 + cc;
      if (chunk < a.nchunk) {
        if (P::SCALED) tile_amax(par);
        __syncthreads();                      // amax complete; every wave has finished reading the previous image
        if (P::SCALED) {
          const int ec = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par]));
          if (ec > E) {
            const float f = pow2f(E - ec);
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[j][r] *= f;
            E = ec;
          }
          par ^= 1;
          if (tid == 0) amax_cell[par] = 0u;
        }
        store_halo(pow2f(14 - E));
        __syncthreads();
        if (chunk + 1 < a.nchunk) load_halo(chunk + 1);
        read_a(toff[0], 0, fa[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          const int cur = (cc * 9 + t) & 1, nxt = cur ^ 1;
          const int T = chunk * 9 + t;
          // step 0
          read_a(toff[t], 1, fa[1]);
          load_b(2 * T + 2, fb[nxt][0]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int m = 0; m < P::NTERM; ++m) acc[j] = P::mfma(fa[0][P::ta(m)], fb[cur][0][j][P::tb(m)], acc[j]);
          __builtin_amdgcn_sched_barrier(0);
          // step 1
          if (t + 1 < 9) read_a(toff[t + 1 < 9 ? t + 1 : 8], 0, fa[0]);
          load_b(2 * T + 3, fb[nxt][1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int m = 0; m < P::NTERM; ++m) acc[j] = P::mfma(fa[1][P::ta(m)], fb[cur][1][j][P::tb(m)], acc[j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __syncthreads();     // all LDS reads done before the stats scratch reuse

  // ---- epilogue (conv_tapset.hip) ----
  float csum[NW], csq[NW];
  float f1 = 1.f, f2 = 1.f;
  if (P::SCALED) {
    const int Ew = exponent_of_bits(*a.ew);
    const int es = E + Ew - 28;
    const bool one = es >= -126 && es <= 127;
    f1 = one ? pow2f(es) : pow2f(E - 14);
    f2 = one ? 1.f : pow2f(Ew - 14);
  }
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    csum[j] = 0.f; csq[j] = 0.f;
    const int n = n0 + 64 * j + 32 * wn + l31;
    if (n >= a.Cd) continue;
    const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const i32x4 pix = *reinterpret_cast<const i32x4*>(&rowpix[32 * wm + 8 * rg + 4 * lh]);
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) {
        const int r = 4 * rg + ri;
        const bool live = pix[ri] >= 0;
        const unsigned e = (unsigned)pix[ri] * (unsigned)a.Cd + (unsigned)n;
        float v = P::SCALED ? fmaf(acc[j][r] * f2, f1, bv) : acc[j][r] + bv;
        if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
        v = live ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
        csum[j] += v; csq[j] += v * v;
      }
    }
  }
  if (a.stats != nullptr) {
    float* red = reinterpret_cast<float*>(smem);     // [wm][64 * NW cols][2]
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

int plane_slots(int ph, int pw) { return (ph + 1) * (pw + 1) + (ph + 1) * pw + ph * (pw + 1) + ph * pw; }

// Ph x Pw <= 64 output pixels with all four planes <= NSMAX slots: fewest tiles, then fewest slots
void s2fwd_patch(int Ho, int Wo, int* Ph, int* Pw) {
  long best = -1;
  *Ph = 1; *Pw = 1;
  for (int pw = 1; pw <= 64 && pw <= Wo + 3; ++pw)
    for (int ph = 1; ph * pw <= 64 && ph <= Ho + 3; ++ph) {
      if (plane_slots(ph, pw) > NSMAX) continue;
      const long tiles = (long)cdiv(Ho, ph) * cdiv(Wo, pw);
      const long cost = tiles * 100000 + plane_slots(ph, pw) * 16 + ((pw & 7) ? 8 : 0);
      if (best < 0 || cost < best) { best = cost; *Ph = ph; *Pw = pw; }
    }
}

template <class P>
int run_s2fwd(F2Args& a, const float* w, void* ws, const unsigned* w_amax, int Cin, int Cout, int nw, hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(w, 9L * Cin * Cout, ws, w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.nchunk * 18 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((conv_s2fwd_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws), a.ew,
                       Cin, Cout, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
  constexpr int lds = P::NPL * PLANE * 2;
  {
    static unsigned long long done[2] = {0ull, 0ull};        // the dynamic-LDS opt-in (above 64 KB) is a per-device function attribute
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
    const int which = nw == 2 ? 1 : 0;
    if (dev < 0 || dev >= 64 || !((done[which] >> dev) & 1ull)) {
      const void* fn = nw == 2 ? reinterpret_cast<const void*>(&conv_s2fwd_kernel<P, 2>) : reinterpret_cast<const void*>(&conv_s2fwd_kernel<P, 1>);
      const hipError_t attr = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (attr != hipSuccess) return (int)attr;
      if (dev >= 0 && dev < 64) done[which] |= 1ull << dev;
    }
  }
  if (nw == 2) hipLaunchKernelGGL((conv_s2fwd_kernel<P, 2>), dim3((unsigned)(a.nx * a.ny)), dim3(256), lds, stream, a);
  else hipLaunchKernelGGL((conv_s2fwd_kernel<P, 1>), dim3((unsigned)(a.nx * a.ny)), dim3(256), lds, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

bool fs_s2fwd_eligible(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil) {
  return R == 3 && S == 3 && stride == 2 && pad == 1 && dil == 1 && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 16 && H >= 2 && W >= 2 &&
         Ho == (H + 1) / 2 && Wo == (W + 1) / 2;
}

long fs_s2fwd_pack_bytes(int mode, int Cin, int Cout) {
  const long nchunk = (Cin + 31) / 32, Npad = ((Cout + 127) / 128) * 128;
  return HDR + nchunk * 18 * (mode == 2 ? 2 : 3) * Npad * 16 * 2;
}

int fs_s2fwd_slabs(int B, int Ho, int Wo) {
  int Ph, Pw;
  s2fwd_patch(Ho, Wo, &Ph, &Pw);
  return B * cdiv(Ho, Ph) * cdiv(Wo, Pw);
}

int fs_s2fwd_conv(int mode, const float* x, const float* w, const float* bias, float* y, float* stats, void* ws, const unsigned* w_amax,
                  int B, int H, int W, int Cin, int Ho, int Wo, int Cout, float drop_scale, uint32_t drop_thresh, uint32_t drop_key,
                  hipStream_t stream) {
  F2Args a;
  a.src = x; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = y; a.stats = stats;
  a.B = B; a.Hs = H; a.Ws = W; a.Cs = Cin; a.Hd = Ho; a.Wd = Wo; a.Cd = Cout;
  a.Npad = ((Cout + 127) / 128) * 128;
  a.nchunk = (Cin + 31) / 32;
  s2fwd_patch(Ho, Wo, &a.Ph, &a.Pw);
  a.tiles_y = cdiv(Ho, a.Ph); a.tiles_x = cdiv(Wo, a.Pw);
  a.base1 = (a.Ph + 1) * (a.Pw + 1);
  a.base2 = a.base1 + (a.Ph + 1) * a.Pw;
  a.base3 = a.base2 + a.Ph * (a.Pw + 1);
  a.nslots = a.base3 + a.Ph * a.Pw;
  if (a.nslots > NSMAX) return FS_ERR_ARG;
  a.magic_pw = div_magic1(a.Pw); a.magic_pw1 = div_magic1(a.Pw + 1);
  a.nx = B * a.tiles_y * a.tiles_x;
  // 128-column workgroups (every plane split once for twice the MFMAs) on layers with >= 128 output channels while the grid still
  // gives the chip ~1.7 workgroups per CU
  static const int nw_pol = FS_ENV_INT("FS_S2FWD_NW", 1);      // kernel A/B builds only: 0 never, 2 whenever Cout >= 128
  const int nw = (nw_pol != 0 && Cout >= 128 && (nw_pol == 2 || (long)a.nx * (a.Npad / 128) >= 440)) ? 2 : 1;
  a.ny = nw == 2 ? a.Npad / 128 : (Cout + 63) / 64;
  const long pack_bytes = fs_s2fwd_pack_bytes(mode, Cin, Cout);
  if (pack_bytes >= 2147483647L || (size_t)B * H * W * Cin * 4 >= 4294967000UL || (size_t)B * Ho * Wo * Cout * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
  a.dst_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * 4);
  a.wp_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  return mode == 2 ? run_s2fwd<PrecF16>(a, w, ws, w_amax, Cin, Cout, nw, stream) : run_s2fwd<PrecX3>(a, w, ws, w_amax, Cin, Cout, nw, stream);
}
