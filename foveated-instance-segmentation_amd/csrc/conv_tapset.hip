// General halo-tiled split-precision convolution (conv_split.h: f16x2 or bf16x3): forward with any stride, and every bwd-data
// sub-problem, written as a list of "tap classes".  A class is a set of filter taps whose source pixels form a
// dense grid in units of `sm` source pixels per output pixel:
//     source row of (output row oy, class tap tr) = sm * (oy + tr) + cy,      filter row r = rbase + rstep * tr
// (same for columns).  Examples: 3x3 stride 1 = one class of 9 taps; 3x3 stride 2 = four classes of 4/2/2/1 taps
// (the input parity planes); 3x3 stride 4 = nine 1-tap classes; 1x1 stride s = one 1-tap class; the bwd-data
// sub-problem of one output parity class = one class with rstep = -stride.
// For each (class, 32-channel chunk) the workgroup loads the class's (Ph+nR-1) x (Pw+nS-1) source halo of its
// Ph x Pw output patch once, (scales and) splits it into 16-bit planes in LDS, and all taps of the class read it at
// shifted slot offsets.  Weights come pre-split from conv_tapset_pack_kernel in consumption order (scaled by the tensor exponent in f16x2; the running
// activation exponent of conv_halo.hip is updated at every LDS refill), as 16-byte
// global loads straight into MFMA B fragments.  Same tiling as conv_halo.hip (128 x 64 tile, wave tile 64 x 32).
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS slot (80 bytes)
constexpr int NSMAX = 224;         // halo slots per plane
constexpr int NITEM = 7;           // NSMAX * 8 quads / 256 threads
constexpr int PLANE = NSMAX * XLD;
struct TsArgs {
  const float* src; const unsigned char* ws; const unsigned* ew; const float* bias; float* dst; float* stats;
  int B, Hs, Ws, Cs, Hd, Wd, Cd;
  int Hq, Wq, os, oy0, ox0, sm;
  int ncls;
  FsTapClass cls[9];
  int Npad, nchunk, ttot, nw;
  int Ph, Pw, tiles_y, tiles_x, nx, ny;
  unsigned src_bytes, wp_bytes, dst_bytes;
  unsigned magic_pw, magic_wh[8];      // div_small1 magics of Pw and of the halo width Pw + nS - 1, nS = 1..8
  float drop_scale; uint32_t drop_thresh, drop_key;
};

// Wp[g = 2*T + s2][plane][n][j]: T enumerates (class, chunk, tap) in consumption order; k = 32*chunk + 16*s2 + j.
//   forward : value = W[r][s][k][n]      bwd-data: value = W[r][s][n][k]      (w is [R][S][Cin][Cout])
template <class P>
__global__ __launch_bounds__(256) void conv_tapset_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew, int Cin, int Cout, int S,
                                                               int transposed, int Ks, int Ns, int Npad, int nchunk, int ncls,
                                                               FsTapClass c0, FsTapClass c1, FsTapClass c2, FsTapClass c3, FsTapClass c4,
                                                               FsTapClass c5, FsTapClass c6, FsTapClass c7, FsTapClass c8, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const FsTapClass cls[9] = {c0, c1, c2, c3, c4, c5, c6, c7, c8};
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int s2 = g & 1;
  int T = g >> 1, r = 0, s = 0, chunk = 0;
  for (int c = 0; c < ncls; ++c) {
    const int nt = cls[c].nR * cls[c].nS;
    if (T < nt * nchunk) {
      chunk = T / nt;
      const int tap = T - chunk * nt;
      const int tr = tap / cls[c].nS, ts = tap - tr * cls[c].nS;
      r = cls[c].rbase + cls[c].rstep * tr;
      s = cls[c].sbase + cls[c].sstep * ts;
      break;
    }
    T -= nt * nchunk;
  }
  const int k0 = chunk * 32 + s2 * 16;
  typename P::x8 p[P::NPL][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Ns && k < Ks) v = transposed ? w[((long)(r * S + s) * Cin + n) * Cout + k] : w[((long)(r * S + s) * Cin + k) * Cout + n];
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

// NW = 32-column sub-tiles per wave (conv_halo.hip): the workgroup covers 64 * NW destination channels, so every class halo is loaded and
// split once for twice the MFMAs when NW = 2.  Round 4 (SQ counters, profiles/r04/pmc/sq_bf16x3_conv_tapset_kernel_fwd_shape7.txt):
// the 64-column form runs 9.2 VALU instructions per MFMA on 64 -> 128 stride 2 with the matrix pipe 37 % busy -- the split / address
// work of a (class, chunk) refill feeds only 24-96 MFMAs per wave -- so it is bound by the vector ALU, not by the matrix cores or HBM.
template <class P, int NW>
__global__ __launch_bounds__(256) void conv_tapset_kernel(TsArgs a) {
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
  const int tpi = a.tiles_y * a.tiles_x;
  const int b = mt / tpi;
  const int trem = mt - b * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * a.Ph, x0 = tx * a.Pw;
  const int npix = a.Ph * a.Pw;

  // output pixel of every tile row (MFMA row order), -1 = dead row
  if (tid < 128) {
    const int p = (tid & ~31) + row_perm(tid & 31);
    const int py = div_small1(p, a.magic_pw), px = p - py * a.Pw;
    const bool live = p < npix && y0 + py < a.Hq && x0 + px < a.Wq;
    rowpix[tid] = live ? ((b * a.Hd + (y0 + py) * a.os + a.oy0) * a.Wd + (x0 + px) * a.os + a.ox0) : -1;
  }
  // A fragment rows of this lane -> patch pixel
  int fpy[2], fpx[2];
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) {
    const int p = 64 * wm + 32 * mi + row_perm(l31);
    const bool live = p < npix;
    fpy[mi] = live ? div_small1(p, a.magic_pw) : 0;
    fpx[mi] = live ? p - fpy[mi] * a.Pw : 0;
  }

  const int q = tid & 7;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.wp_bytes);
  if (tid < 2) amax_cell[tid] = 0u;

  // ---- halo loader state (runs one (class, chunk) ahead of the MFMA loop) ----
  int goff[NITEM];
  f32x4 ra[NITEM];
  int pc = 0, pchunk = 0;        // (class, chunk) the registers ra hold / will hold
  auto class_offsets = [&](int c) {
    const int Wh = a.Pw + a.cls[c].nS - 1, nslots = (a.Ph + a.cls[c].nR - 1) * Wh;
    const int cy = a.cls[c].cy, cx = a.cls[c].cx;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (tid >> 3) + 32 * i;
      goff[i] = -1;
      if (slot < nslots) {
        const int hy = div_small1(slot, a.magic_wh[a.cls[c].nS - 1]), hx = slot - hy * Wh;
        const int sy = a.sm * (y0 + hy) + cy, sx = a.sm * (x0 + hx) + cx;
        if (sy >= 0 && sy < a.Hs && sx >= 0 && sx < a.Ws) goff[i] = ((b * a.Hs + sy) * a.Ws + sx) * a.Cs + 4 * q;
      }
    }
  };
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

  // ---- B fragments ----
  const int bvoff = HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = 2 * a.ttot;
  auto load_b = [&](int g, X8 (&dst)[NW][NPL]) {
    const int gg = g < G ? g : G - 1;
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, bvoff + j * 2048, gg * step_bytes + pl * plane_bytes, 0);      // sub-tile j: + 64 columns
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
  X8 fa[2][2][NPL];         // [k16 step][mi][plane]
  X8 fbA[2][NW][NPL], fbB[2][NW][NPL];
  int E = EMIN, par = 0;
  // MFMA-loop state
  int c = 0, chunk = 0, tap = 0, tr = 0, ts = 0;
  int ntaps = a.cls[0].nR * a.cls[0].nS, nS = a.cls[0].nS, Wh = a.Pw + a.cls[0].nS - 1;
  int rowbase[2];
  int T = 0;

  auto read_a = [&](int toff, int s2, X8 (&dst)[2][NPL]) {
#pragma unroll
    for (int mi = 0; mi < 2; ++mi)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) dst[mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase[mi] + toff + 16 * s2]);
  };
  auto mfmas = [&](const X8 (&A)[2][NPL], const X8 (&Bf)[NW][NPL]) {
#pragma unroll
    for (int j = 0; j < NW; ++j)
#pragma unroll
      for (int t = 0; t < P::NTERM; ++t) {           // smallest terms first, the two pixel halves interleaved
        acc[0][j] = P::mfma(A[0][P::ta(t)], Bf[j][P::tb(t)], acc[0][j]);
        acc[1][j] = P::mfma(A[1][P::ta(t)], Bf[j][P::tb(t)], acc[1][j]);
      }
  };
  // one filter tap = two k16 steps; `cur` holds this tap's B fragments, `nxt` receives the next tap's
  auto tap_body = [&](X8 (&cur)[2][NW][NPL], X8 (&nxt)[2][NW][NPL]) {
    if (tap == 0) {                      // first tap of a (class, chunk): refill LDS
      if (chunk == 0) {
        ntaps = a.cls[c].nR * a.cls[c].nS; nS = a.cls[c].nS; Wh = a.Pw + nS - 1;
#pragma unroll
        for (int mi = 0; mi < 2; ++mi) rowbase[mi] = (fpy[mi] * Wh + fpx[mi]) * XLD + 8 * lh;
      }
      if (P::SCALED) tile_amax(par);
      __syncthreads();                      // amax complete; every wave has finished reading the previous image
      if (P::SCALED) {
        const int ec = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par]));
        if (ec > E) {
          const float f = pow2f(E - ec);
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[mi][j][r] *= f;
          E = ec;
        }
        par ^= 1;
        if (tid == 0) amax_cell[par] = 0u;
      }
      store_halo(pow2f(14 - E));
      __syncthreads();
      if (++pchunk == a.nchunk) { pchunk = 0; ++pc; if (pc < a.ncls) class_offsets(pc); }
      if (pc < a.ncls) load_halo(pchunk);
      tr = 0; ts = 0;
      read_a(0, 0, fa[0]);
    }
    const int toff = (tr * Wh + ts) * XLD;
    // step 0
    read_a(toff, 1, fa[1]);
    load_b(2 * T + 2, nxt[0]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[0], cur[0]);
    __builtin_amdgcn_sched_barrier(0);
    // step 1
    int ntr = tr, nts = ts + 1;
    if (nts == nS) { nts = 0; ++ntr; }
    if (tap + 1 < ntaps) read_a((ntr * Wh + nts) * XLD, 0, fa[0]);
    load_b(2 * T + 3, nxt[1]);
    __builtin_amdgcn_sched_barrier(0);
    mfmas(fa[1], cur[1]);
    __builtin_amdgcn_sched_barrier(0);
    tr = ntr; ts = nts;
    ++T;
    if (++tap == ntaps) { tap = 0; if (++chunk == a.nchunk) { chunk = 0; ++c; } }
  };

  if (a.ttot > 0) {
    load_b(0, fbA[0]);
    load_b(1, fbA[1]);
    class_offsets(0);
    load_halo(0);
    __syncthreads();                        // amax cells zeroed before the first atomic
    while (T < a.ttot) {
      tap_body(fbA, fbB);
      if (T < a.ttot) tap_body(fbB, fbA);
    }
  }
  __syncthreads();     // rowpix visible (and all LDS reads done before the stats scratch reuse)

  // ---- epilogue ----
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
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int rg = 0; rg < 4; ++rg) {
        const i32x4 pix = *reinterpret_cast<const i32x4*>(&rowpix[64 * wm + 32 * mi + 8 * rg + 4 * lh]);
#pragma unroll
        for (int ri = 0; ri < 4; ++ri) {
          const int r = 4 * rg + ri;
          const bool live = pix[ri] >= 0;
          const unsigned e = (unsigned)pix[ri] * (unsigned)a.Cd + (unsigned)n;
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
    __syncthreads();
    float* red = reinterpret_cast<float*>(&Ah[0]);     // [wm][64 * NW cols][2]
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

// Ph x Pw <= 128 output pixels with every class halo (Ph+nR-1)(Pw+nS-1) <= NSMAX; fewest tiles, then smallest halo.
void fs_tapset_patch(int Hq, int Wq, int maxR, int maxS, int* Ph, int* Pw) {
  long best = -1;
  *Ph = 1; *Pw = 1;
  for (int pw = 1; pw <= 128 && pw <= Wq + 3; ++pw) {
    int ph = 128 / pw;
    if (ph > Hq) ph = Hq;
    while (ph > 1 && (ph + maxR - 1) * (pw + maxS - 1) > NSMAX) --ph;
    if (ph < 1 || (ph + maxR - 1) * (pw + maxS - 1) > NSMAX) continue;
    const long tiles = (long)cdiv(Hq, ph) * cdiv(Wq, pw);
    const long cost = tiles * 100000 + (ph + maxR - 1) * (pw + maxS - 1) * 16 + ((pw & 15) ? 8 : 0);
    if (best < 0 || cost < best) { best = cost; *Ph = ph; *Pw = pw; }
  }
}

long fs_tapset_pack_bytes(int mode, int Cs, int Cd, int total_taps) {
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 127) / 128) * 128;       // room for either column tiling
  return HDR + nchunk * total_taps * 2 * (mode == 2 ? 2 : 3) * Npad * 16 * 2;
}

int fs_tapset_slabs(int B, int Hq, int Wq, int maxR, int maxS) {
  int Ph, Pw;
  fs_tapset_patch(Hq, Wq, maxR, maxS, &Ph, &Pw);
  return B * cdiv(Hq, Ph) * cdiv(Wq, Pw);
}

namespace {
template <class P>
int run_tapset(TsArgs& a, const FsTapsetProblem& p, hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(p.w, (long)p.R * p.S * p.Cin * p.Cout, p.ws, p.w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.ttot * 2 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((conv_tapset_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, p.w,
                       reinterpret_cast<unsigned char*>(p.ws), a.ew, p.Cin, p.Cout, p.S, p.transposed, p.Cs, p.Cd, a.Npad, a.nchunk, a.ncls,
                       a.cls[0], a.cls[1], a.cls[2], a.cls[3], a.cls[4], a.cls[5], a.cls[6], a.cls[7], a.cls[8], total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
#ifdef FS_EXPERIMENTS          // the 128-column form is measured-and-rejected (below): only the A/B build carries it
  if (a.nw == 2) hipLaunchKernelGGL((conv_tapset_kernel<P, 2>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  else
#endif
  hipLaunchKernelGGL((conv_tapset_kernel<P, 1>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
}  // namespace

int fs_tapset_conv(int mode, const FsTapsetProblem& p, hipStream_t stream) {
  TsArgs a;
  a.src = p.src; a.ws = reinterpret_cast<const unsigned char*>(p.ws); a.bias = p.bias; a.dst = p.dst; a.stats = p.stats;
  a.B = p.B; a.Hs = p.Hs; a.Ws = p.Ws; a.Cs = p.Cs; a.Hd = p.Hd; a.Wd = p.Wd; a.Cd = p.Cd;
  a.Hq = p.Hq; a.Wq = p.Wq; a.os = p.os; a.oy0 = p.oy0; a.ox0 = p.ox0; a.sm = p.sm;
  a.ncls = p.ncls;
  int maxR = 1, maxS = 1, total_taps = 0;
  for (int c = 0; c < 9; ++c) {
    a.cls[c] = c < p.ncls ? p.cls[c] : FsTapClass{0, 0, 0, 0, 0, 0, 0, 0};
    if (c < p.ncls) {
      if (p.cls[c].nR < 1 || p.cls[c].nS < 1) return FS_ERR_ARG;
      if (p.cls[c].nR > maxR) maxR = p.cls[c].nR;
      if (p.cls[c].nS > maxS) maxS = p.cls[c].nS;
      total_taps += p.cls[c].nR * p.cls[c].nS;
    }
  }
  if (p.ncls < 1 || p.ncls > 9 || total_taps < 1) return FS_ERR_ARG;
  a.Npad = ((p.Cd + 127) / 128) * 128;
  a.nchunk = (p.Cs + 31) / 32;
  a.ttot = a.nchunk * total_taps;
  fs_tapset_patch(p.Hq, p.Wq, maxR, maxS, &a.Ph, &a.Pw);
  if ((a.Ph + maxR - 1) * (a.Pw + maxS - 1) > NSMAX) return FS_ERR_ARG;
  a.tiles_y = cdiv(p.Hq, a.Ph); a.tiles_x = cdiv(p.Wq, a.Pw);
  a.magic_pw = div_magic1(a.Pw);
  if (maxS > 8) return FS_ERR_ARG;
  for (int ns = 1; ns <= 8; ++ns) a.magic_wh[ns - 1] = div_magic1(a.Pw + ns - 1);
  a.nx = p.B * a.tiles_y * a.tiles_x;
  // 128-column workgroups (two sub-tiles per wave: every class halo split once for twice the MFMAs) were measured against the 64-column
  // form in one gpurun call (profiles/r04/tapset_nw_ab.txt, us, bf16x3, B = 64): forward 64 -> 128 s2 @ 80x80 145 -> 159, 128 -> 256 @ 40x40
  // 151 -> 152, 256 -> 512 @ 20x20 146 -> 143-151, 512 -> 512 @ 20x20 277 -> 279; bwd-data the same or slower.  Every strided shape sits at
  // ~100 TF whatever its K: what bounds the kernel is the number of (class, chunk) LDS refills -- two barriers and a split phase around
  // 24-96 MFMAs per wave -- not the split work per MFMA.  Policy 0 (shipped) = 64 columns; 1 / 2 only in the A/B build.
  static const int nw_pol = FS_ENV_INT("FS_TAPSET_NW", 0);
  a.nw = (nw_pol != 0 && p.Cd >= 128 && (nw_pol == 2 || (long)a.nx * (a.Npad / 128) >= 440)) ? 2 : 1;
  a.ny = a.nw == 2 ? a.Npad / 128 : (p.Cd + 63) / 64;
  const long pack_bytes = fs_tapset_pack_bytes(mode, p.Cs, p.Cd, total_taps);
  if (pack_bytes >= 2147483647L || (size_t)p.B * p.Hs * p.Ws * p.Cs * 4 >= 4294967000UL || (size_t)p.B * p.Hd * p.Wd * p.Cd * 4 >= 4294967000UL)
    return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)p.B * p.Hs * p.Ws * p.Cs * 4);
  a.dst_bytes = (unsigned)((size_t)p.B * p.Hd * p.Wd * p.Cd * 4);
  a.wp_bytes = (unsigned)pack_bytes;
  a.drop_scale = p.drop_scale; a.drop_thresh = p.drop_thresh; a.drop_key = p.drop_key;
  return mode == 2 ? run_tapset<PrecF16>(a, p, stream) : run_tapset<PrecX3>(a, p, stream);
}
