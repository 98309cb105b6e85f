// bwd-data of a 3x3 / stride 2 / pad 1 convolution in ONE launch (round 4; split precision of conv_split.h: f16x2 or bf16x3).
//
// dX row y = 2 q + a receives filter row r iff (y + 1 - r) is even, from dY row (y + 1 - r) / 2:
//     a = 0 :  r = 1 from dY row q                 a = 1 :  r = 2 from dY row q,  r = 0 from dY row q + 1          (columns alike)
// so the four output parities (a, b) of the dX pixels 2 q + (a, b) are four small convolutions over the SAME (Ph + 1) x (Pw + 1) dY halo
// of a Ph x Pw patch of q, with 1 / 2 / 2 / 4 taps.  The tap-class kernel (conv_tapset.hip) runs them as four launches, each with its
// own weight pack, LDS refills around 24-96 MFMAs per wave and a grid of 1.6 rounds of workgroups: 56-88 TF on the HRNet fuse
// down-paths where the stride-1 kernels reach 230 (profiles/r04/shape_table_base.txt).  Here a workgroup (4 waves: 2 pixel halves x 2
// channel halves of a 64-pixel x 64-channel tile) keeps FOUR accumulators per wave, one per parity, loads and splits the halo once per
// 32-channel chunk and runs all nine (parity, tap) products on it: 108 MFMAs per wave and refill in bf16x3, one pack kernel, one
// launch of B * tiles * (Cin / 64) workgroups.  Weights come pre-split in consumption order (chunk, product, k16 step) as 16-byte
// global loads straight into the B operand registers, as in the other halo-tiled kernels.
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS slot (80 bytes)
constexpr int NSMAX = 96;          // halo slots per plane: (Ph + 1)(Pw + 1) for Ph x Pw <= 64
constexpr int NITEM = 3;           // NSMAX * 8 quads / 256 threads
constexpr int PLANE = NSMAX * XLD;

// the nine products: parity class c = 2 a + b, filter tap (r, s), halo offset (dyo, dxo) of the dY pixel relative to q
struct Prod { int c, r, s, dyo, dxo; };
__host__ __device__ constexpr Prod prod_of(int t) {
  return t == 0 ? Prod{0, 1, 1, 0, 0}
       : t == 1 ? Prod{1, 1, 2, 0, 0} : t == 2 ? Prod{1, 1, 0, 0, 1}
       : t == 3 ? Prod{2, 2, 1, 0, 0} : t == 4 ? Prod{2, 0, 1, 1, 0}
       : t == 5 ? Prod{3, 2, 2, 0, 0} : t == 6 ? Prod{3, 2, 0, 0, 1} : t == 7 ? Prod{3, 0, 2, 1, 0} : Prod{3, 0, 0, 1, 1};
}

struct S2Args {
  const float* src; const unsigned char* ws; const unsigned* ew; float* dst;
  int B, Hs, Ws, Cs, Hd, Wd, Cd;         // src = dY (B,Hs,Ws,Cs = Cout), dst = dX (B,Hd,Wd,Cd = Cin)
  int Npad, nchunk;
  int Ph, Pw, tiles_y, tiles_x, nx, ny;
  unsigned src_bytes, wp_bytes, dst_bytes;
  unsigned magic_pw, magic_wh;
  // extras (round 5): BatchNorm-backward column sums of the layer whose output gradient dX is (stats[tile][Cd][2] = sum dz, sum dz * zhat,
  // dz = dX masked by that layer's activation bits) and a second gradient added to dX in the same epilogue
  const float* bn_y; const unsigned char* bn_mask; const float* bn_mean; const float* bn_invstd;
  const float* add_src; const unsigned char* add_mask; float* stats;
};

// Wp[g = 2 * (chunk * 9 + t) + s2][plane][n][j] = plane-th term of W[r_t][s_t][n][k = 32 chunk + 16 s2 + j]   (w is [R][S][Cin][Cout]:
// N = Cin, K = Cout), scaled by 2^(14 - Ew) in f16x2, behind a HDR-byte header
template <class P>
__global__ __launch_bounds__(256) void conv_s2bwd_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew,
                                                              int Cin, int Cout, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int s2 = g & 1, T = g >> 1;
  const int chunk = T / 9, t = T - chunk * 9;
  int r = 0, s = 0;
#pragma unroll
  for (int k = 0; k < 9; ++k) if (k == t) { r = prod_of(k).r; s = prod_of(k).s; }
  const int k0 = chunk * 32 + s2 * 16;
  typename P::x8 p[P::NPL][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Cin && k < Cout) v = w[((long)(r * 3 + s) * Cin + n) * Cout + k];
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

template <class P, bool BN = false>
__global__ __launch_bounds__(256, 2) void conv_s2bwd_kernel(S2Args a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  __shared__ __attribute__((aligned(16))) typename P::T Ah[NPL * PLANE];
  __shared__ __attribute__((aligned(16))) int rowpix[64];      // dX pixel index of parity (0, 0) of every tile row, -1 = dead row
  __shared__ __attribute__((aligned(16))) int rowflag[64];     // bit 0: dX row 2 q + 1 exists, bit 1: dX column 2 q + 1 exists
  __shared__ unsigned amax_cell[2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;      // XCD-aware: the channel tiles of a patch share an L2
  const int mt = wg / a.ny;
  const int n0 = (wg - mt * a.ny) * 64;
  const int tpi = a.tiles_y * a.tiles_x;
  const int b = mt / tpi;
  const int trem = mt - b * tpi;
  const int ty = trem / a.tiles_x, tx = trem - ty * a.tiles_x;
  const int y0 = ty * a.Ph, x0 = tx * a.Pw;
  const int npix = a.Ph * a.Pw, Wh = a.Pw + 1, nslots = (a.Ph + 1) * Wh;

  if (tid < 64) {
    const int p = (tid & ~31) + row_perm(tid & 31);
    const int py = div_small1(p, a.magic_pw), px = p - py * a.Pw;
    const int yy = 2 * (y0 + py), xx = 2 * (x0 + px);
    const bool live = p < npix && yy < a.Hd && xx < a.Wd;
    rowpix[tid] = live ? (b * a.Hd + yy) * a.Wd + xx : -1;
    rowflag[tid] = (yy + 1 < a.Hd ? 1 : 0) | (xx + 1 < a.Wd ? 2 : 0);
  }
  // A fragment row of this lane -> patch pixel -> LDS element offset of its halo slot (dyo = dxo = 0)
  int rowbase;
  {
    const int p = 32 * wm + row_perm(l31);
    const bool live = p < npix;
    const int py = live ? div_small1(p, a.magic_pw) : 0, px = live ? p - py * a.Pw : 0;
    rowbase = (py * Wh + px) * XLD + 8 * lh;
  }

  const int q = tid & 7;
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.wp_bytes);
  if (tid < 2) amax_cell[tid] = 0u;

  // ---- halo loader: slot (tid >> 3) + 32 i, channel quad q; one chunk ahead of the MFMA loop ----
  int goff[NITEM];
  f32x4 ra[NITEM];
#pragma unroll
  for (int i = 0; i < NITEM; ++i) {
    const int slot = (tid >> 3) + 32 * i;
    goff[i] = -1;
    if (slot < nslots) {
      const int hy = div_small1(slot, a.magic_wh), hx = slot - hy * Wh;
      const int sy = y0 + hy, sx = x0 + hx;
      if (sy < a.Hs && sx < a.Ws) goff[i] = ((b * a.Hs + sy) * a.Ws + sx) * a.Cs + 4 * q;
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

  // ---- B fragments: step g = 2 * (chunk * 9 + t) + s2 ----
  const int bvoff = HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = 18 * a.nchunk;
  auto load_b = [&](int g, X8 (&dst)[NPL]) {
    const int gg = g < G ? g : G - 1;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, bvoff, gg * step_bytes + pl * plane_bytes, 0);
      dst[pl] = __builtin_bit_cast(X8, v);
    }
  };

  f32x16 acc[4];
#pragma unroll
  for (int c = 0; c < 4; ++c)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
  X8 fa[2][NPL];            // [k16 step][plane]
  X8 fb[2][2][NPL];         // [product parity][k16 step][plane]: the next product's fragments load while this one multiplies
  int E = EMIN, par = 0;

  auto read_a = [&](int toff, int s2, X8 (&dst)[NPL]) {
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) dst[pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase + toff + 16 * s2]);
  };

  load_b(0, fb[0][0]);
  load_b(1, fb[0][1]);
  load_halo(0);
  __syncthreads();                        // amax cells zeroed before the first atomic; rowpix / rowflag written
  // two chunks per trip: 9 products per chunk is odd, so the fragment buffer of a product alternates from chunk to chunk
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
            for (int c = 0; c < 4; ++c)
#pragma unroll
              for (int r = 0; r < 16; ++r) acc[c][r] *= f;
            E = ec;
          }
          par ^= 1;
          if (tid == 0) amax_cell[par] = 0u;
        }
        store_halo(pow2f(14 - E));
        __syncthreads();
        if (chunk + 1 < a.nchunk) load_halo(chunk + 1);
        read_a((prod_of(0).dyo * Wh + prod_of(0).dxo) * XLD, 0, fa[0]);
#pragma unroll
        for (int t = 0; t < 9; ++t) {
          constexpr int dummy = 0; (void)dummy;
          const int cur = (cc * 9 + t) & 1, nxt = cur ^ 1;
          const int T = chunk * 9 + t;
          const int toff = (prod_of(t).dyo * Wh + prod_of(t).dxo) * XLD;
          // step 0
          read_a(toff, 1, fa[1]);
          load_b(2 * T + 2, fb[nxt][0]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < P::NTERM; ++m) acc[prod_of(t).c] = P::mfma(fa[0][P::ta(m)], fb[cur][0][P::tb(m)], acc[prod_of(t).c]);
          __builtin_amdgcn_sched_barrier(0);
          // step 1
          if (t + 1 < 9) read_a((prod_of(t + 1 < 9 ? t + 1 : 8).dyo * Wh + prod_of(t + 1 < 9 ? t + 1 : 8).dxo) * XLD, 0, fa[0]);
          load_b(2 * T + 3, fb[nxt][1]);
          __builtin_amdgcn_sched_barrier(0);
#pragma unroll
          for (int m = 0; m < P::NTERM; ++m) acc[prod_of(t).c] = P::mfma(fa[1][P::ta(m)], fb[cur][1][P::tb(m)], acc[prod_of(t).c]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  __syncthreads();

  // ---- epilogue: parity (a_, b_) of tile row p goes to dX pixel rowpix[p] + a_ * Wd + b_ ----
  const int n = n0 + 32 * wn + l31;
  if (n < a.Cd) {
    float f1 = 1.f, f2 = 1.f;
    if (P::SCALED) {
      const int Ew = exponent_of_bits(*a.ew);
      const int es = E + Ew - 28;
      const bool one = es >= -126 && es <= 127;
      f1 = one ? pow2f(es) : pow2f(E - 14);
      f2 = one ? 1.f : pow2f(Ew - 14);
    }
    const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
    if constexpr (!BN) {
#pragma unroll
    for (int rg = 0; rg < 4; ++rg) {
      const i32x4 pix = *reinterpret_cast<const i32x4*>(&rowpix[32 * wm + 8 * rg + 4 * lh]);
      const i32x4 flg = *reinterpret_cast<const i32x4*>(&rowflag[32 * wm + 8 * rg + 4 * lh]);
#pragma unroll
      for (int ri = 0; ri < 4; ++ri) {
        const int r = 4 * rg + ri;
#pragma unroll
        for (int c = 0; c < 4; ++c) {
          const int a_ = c >> 1, b_ = c & 1;
          const bool live = pix[ri] >= 0 && (a_ == 0 || (flg[ri] & 1)) && (b_ == 0 || (flg[ri] & 2));
          const unsigned e = (unsigned)(pix[ri] + a_ * a.Wd + b_) * (unsigned)a.Cd + (unsigned)n;
          const float v = P::SCALED ? acc[c][r] * f2 * f1 : acc[c][r];
          __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
        }
      }
    }
    }
  }
  if constexpr (BN) {
    // lane = column, (tile row, parity) = register: the extras' operands sit at the addresses of the stores.  One group of two tile rows x
    // four parities is requested while the group before it is stored; an absent operand gets a zero-size descriptor (loads return 0).
    const bool nok = n < a.Cd;
    float f1 = 1.f, f2 = 1.f;
    if (P::SCALED) {
      const int Ew = exponent_of_bits(*a.ew);
      const int es = E + Ew - 28;
      const bool one = es >= -126 && es <= 127;
      f1 = one ? pow2f(es) : pow2f(E - 14);
      f2 = one ? 1.f : pow2f(Ew - 14);
    }
    const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
    const bool want_y = a.bn_y != nullptr, want_a = a.add_src != nullptr;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    const float mu = (want_y && nok) ? a.bn_mean[n] : 0.f, is = (want_y && nok) ? a.bn_invstd[n] : 0.f;
    const unsigned bit = (unsigned)(n & 3);
    float yv[2][8], av[2][8];
    unsigned ymv[2][8], amv[2][8];
    auto element = [&](int g, int k, unsigned& e) -> bool {          // group g = (row group rg, row pair): k = 4 (ri & 1) + parity c
      const int rg = g >> 1, ri = 2 * (g & 1) + (k >> 2), c = k & 3, a_ = c >> 1, b_ = c & 1;
      const int row = 32 * wm + 8 * rg + 4 * lh + ri;
      const int pix = rowpix[row], flg = rowflag[row];
      e = (unsigned)(pix + a_ * a.Wd + b_) * (unsigned)a.Cd + (unsigned)n;
      return nok && pix >= 0 && (a_ == 0 || (flg & 1)) && (b_ == 0 || (flg & 2));
    };
    auto issue = [&](int g, int slot) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned e;
        const bool live = element(g, k, e);
        const int o4 = live ? (int)(e * 4u) : (int)OOB, o1 = live ? (int)(e >> 2) : (int)OOB;
        yv[slot][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_y, o4, 0, 0));
        ymv[slot][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_ym, o1, 0, 0) | ym_all;
        av[slot][k] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs_a, o4, 0, 0));
        amv[slot][k] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_am, o1, 0, 0) | am_all;
      }
    };
    float csum = 0.f, csq = 0.f;
    issue(0, 0);
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const int slot = g & 1;
      if (g + 1 < 8) issue(g + 1, slot ^ 1);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = 4 * (g >> 1) + 2 * (g & 1) + (k >> 2), c = k & 3;
        unsigned e;
        const bool live = element(g, k, e);
        float v = P::SCALED ? acc[c][r] * f2 * f1 : acc[c][r];
        if (want_a) v += ((amv[slot][k] >> bit) & 1u) ? av[slot][k] : 0.f;
        v = live ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
        const float dz = ((ymv[slot][k] >> bit) & 1u) ? v : 0.f;
        csum += dz; csq += dz * ((yv[slot][k] - mu) * is);
      }
    }
    if (a.stats != nullptr) {
      // the wave's 32 tile rows: lanes l and l ^ 32 hold the two row halves of a column; the two pixel halves (wm) meet in LDS
      csum += __shfl_xor(csum, 32, 64); csq += __shfl_xor(csq, 32, 64);
      float* red = reinterpret_cast<float*>(&Ah[0]);       // [wm][64 columns][2]; the halo image is dead since the barrier behind the loop
      if (lh == 0) { red[(wm * 64 + 32 * wn + l31) * 2] = csum; red[(wm * 64 + 32 * wn + l31) * 2 + 1] = csq; }
      __syncthreads();
      if (tid < 128) {
        const int col = tid >> 1, which = tid & 1;
        if (n0 + col < a.Cd) a.stats[((long)mt * a.Cd + n0 + col) * 2 + which] = red[col * 2 + which] + red[(64 + col) * 2 + which];
      }
    }
  }
}

// Ph x Pw <= 64 pixels of the q grid with halo (Ph + 1)(Pw + 1) <= NSMAX: fewest tiles, then smallest halo
void s2bwd_patch(int Hq, int Wq, int* Ph, int* Pw) {
  long best = -1;
  *Ph = 1; *Pw = 1;
  for (int pw = 1; pw <= 64 && pw <= Wq + 3; ++pw)
    for (int ph = 1; ph * pw <= 64 && ph <= Hq + 3; ++ph) {
      if ((ph + 1) * (pw + 1) > NSMAX) continue;
      const long tiles = (long)cdiv(Hq, ph) * cdiv(Wq, pw);
      const long cost = tiles * 100000 + (ph + 1) * (pw + 1) * 16 + ((pw & 7) ? 8 : 0);
      if (best < 0 || cost < best) { best = cost; *Ph = ph; *Pw = pw; }
    }
}

template <class P>
int run_s2bwd(S2Args& a, const float* w, void* ws, const unsigned* w_amax, int Cin, int Cout, hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(w, 9L * Cin * Cout, ws, w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.nchunk * 18 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((conv_s2bwd_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws), a.ew,
                       Cin, Cout, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
  if (a.bn_y != nullptr || a.add_src != nullptr) hipLaunchKernelGGL((conv_s2bwd_kernel<P, true>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv_s2bwd_kernel<P, false>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

bool fs_s2bwd_eligible(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil) {
  return R == 3 && S == 3 && stride == 2 && pad == 1 && dil == 1 && Cin % 4 == 0 && Cout % 4 == 0 && Cout >= 16 && H >= 2 && W >= 2 &&
         Ho == (H + 1) / 2 && Wo == (W + 1) / 2;
}

long fs_s2bwd_pack_bytes(int mode, int Cin, int Cout) {
  const long nchunk = (Cout + 31) / 32, Npad = ((Cin + 63) / 64) * 64;
  return HDR + nchunk * 18 * (mode == 2 ? 2 : 3) * Npad * 16 * 2;
}

// dX (B,H,W,Cin) of a 3x3 / stride 2 / pad 1 convolution from dY (B,Ho,Wo,Cout); every element of dX is written
// rows of the [rows][Cin][2] slab the extras' column sums go to: one per patch of the q grid
int fs_s2bwd_stats_slabs(int B, int Ho, int Wo) {
  int Ph, Pw;
  s2bwd_patch(Ho, Wo, &Ph, &Pw);
  return B * cdiv(Ho, Ph) * cdiv(Wo, Pw);
}

int fs_s2bwd_conv(int mode, const float* dy, const float* w, float* dx, void* ws, const unsigned* w_amax, int B, int H, int W, int Cin, int Ho,
                  int Wo, int Cout, const FsBnSums* bn, float* slab, hipStream_t stream) {
  S2Args a;
  if (bn != nullptr && bn->y != nullptr && slab == nullptr) return FS_ERR_ARG;
  a.bn_y = bn ? bn->y : nullptr; a.bn_mask = bn ? bn->mask : nullptr; a.bn_mean = bn ? bn->mean : nullptr; a.bn_invstd = bn ? bn->invstd : nullptr;
  a.add_src = bn ? bn->add_src : nullptr; a.add_mask = bn ? bn->add_mask : nullptr; a.stats = (bn != nullptr && bn->y != nullptr) ? slab : nullptr;
  a.src = dy; a.ws = reinterpret_cast<const unsigned char*>(ws); a.dst = dx;
  a.B = B; a.Hs = Ho; a.Ws = Wo; a.Cs = Cout; a.Hd = H; a.Wd = W; a.Cd = Cin;
  a.Npad = ((Cin + 63) / 64) * 64;
  a.nchunk = (Cout + 31) / 32;
  s2bwd_patch(Ho, Wo, &a.Ph, &a.Pw);
  a.tiles_y = cdiv(Ho, a.Ph); a.tiles_x = cdiv(Wo, a.Pw);
  a.magic_pw = div_magic1(a.Pw); a.magic_wh = div_magic1(a.Pw + 1);
  a.nx = B * a.tiles_y * a.tiles_x;
  a.ny = a.Npad / 64;
  const long pack_bytes = fs_s2bwd_pack_bytes(mode, Cin, Cout);
  if (pack_bytes >= 2147483647L || (size_t)B * Ho * Wo * Cout * 4 >= 4294967000UL || (size_t)B * H * W * Cin * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * 4);
  a.dst_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
  a.wp_bytes = (unsigned)pack_bytes;
  return mode == 2 ? run_s2bwd<PrecF16>(a, w, ws, w_amax, Cin, Cout, stream) : run_s2bwd<PrecX3>(a, w, ws, w_amax, Cin, Cout, stream);
}
