// 3x3 / stride-1 / pad-1 convolution (forward and bwd-data) with the minimal-filtering identity F(2,3) applied along the image
// row: two neighbouring output columns (a "pair") are computed from 4 transform-domain products instead of 6 taps, so a 3x3
// filter costs 3 (rows) x 4 = 12 MFMA steps per pair of pixels where conv_halo.hip spends 18.
//
//   inputs of pair j, row y:  d0..d3 = x[y][2j-1 .. 2j+2]          T0 = d0 - d2   T1 = d1 + d2   T2 = d2 - d1   T3 = d1 - d3
//   filter row (g0 g1 g2):                                         U0 = g0   U1 = (g0+g1+g2)/2   U2 = (g0-g1+g2)/2   U3 = g2
//   m_c = sum over the 3 filter rows and the input channels of T_c * U_c        (4 independent GEMMs, fp32 accumulate)
//   out[2j] = m0 + m1 + m2        out[2j+1] = m1 - m2 - m3
//
// The transform is exact in the same sense the rest of the path is: T and U are formed in fp32 (one rounding each), THEN split
// into the 16-bit planes of the precision mode (conv_split.h), so the products carry the mode's operand width; the error
// against fp64 is measured per layer by tools/conv_accuracy.py (profiles/r02/conv_accuracy_winograd.txt).
//
//   workgroup = 256 threads = 4 waves as 2 (component pairs {0,1} / {2,3}) x 2 (32-column halves); it owns 64 pairs
//   (Ph rows x PP pairs = 128 output pixels) x 64 output channels.  A wave holds 2 components x 64 pairs x 32 columns.
//   LDS: NPL planes x 4 components x 80 halo slots x 80 B (32 k + 16 B pad), dynamic (76.8 KB in bf16x3, two workgroups per CU).
//   The batch is tiled as ONE image of B*(H+1) rows (a zero row after every image: the vertical padding neighbours share).
//   Weights are transformed and split ahead of time by a pack kernel into the order each wave consumes its B fragments.
//   Workgroups are PERSISTENT: 2 per CU, each walks a strided list of tiles of its XCD's contiguous tile range; the first halo image
//   and the first weight fragments of the next tile are requested while the current tile's rows are being stored (round 3: the
//   first-load latency was 3.9 k of the 35 k cycles a 64-channel tile takes).
//   Epilogue: every wave writes its two partial inverse-transform sums (even pixel / odd pixel) to LDS in accumulator order and the
//   wave that owns a pixel parity reads them back as [pixel][4 channels] rows: 16-byte global stores (8 per wave instead of 64 dword
//   stores; round 3), bias / dropout / BatchNorm partial sums on those rows.
#include "conv_split.h"
#include "conv_kernels.h"
#include "conv_wino_diag.h"

#include <cstdio>
#include <cstdlib>

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS slot (80 bytes)
constexpr int WNS = 80;            // halo slots per component image: (Ph + 2) * PP <= WNS
constexpr int CPLANE = WNS * XLD;  // elements per component image
constexpr int PLANE = 4 * CPLANE;  // elements per precision plane

struct WinoArgs {
  const float* src; const unsigned char* ws; const unsigned* ew; const float* bias; float* dst; float* stats;
  int B, H, W, Cs, Cd, Npad, nchunk;
  int Ph, PP, tiles_x, nx, ny, Hv;
  unsigned src_bytes, ws_bytes, dst_bytes;
  unsigned magic_pp, magic_hv, magic_ny, magic_tx;
  // bwd-data only: dst is the gradient dz of a conv + BatchNorm + activation layer's output -> `stats` receives that layer's
  // BatchNorm-backward column sums (sum g, sum g * xhat; g = dz * act', xhat = (y - mean) * invstd) instead of (sum v, sum v^2)
  const float* bn_y; const unsigned char* bn_mask; const float* bn_mean; const float* bn_invstd;
  // ... and (optionally) dst = conv result + add_src [masked by add_mask: 1 byte per 4 channels]: the residual branch's gradient joins the
  // conv branch's in this epilogue instead of in a separate n-ary add (add_src = the next BatchNorm's dz, add_mask = its activation bits)
  const float* add_src; const unsigned char* add_mask;
  // forward, inference: dst = act((conv + bias) * ep_scale[c] + ep_shift[c] [+ ep_res]) -- eval-mode BatchNorm, residual and activation in
  // the epilogue, so the separate normalise / activate pass over the conv output disappears (ep_scale NULL = off)
  const float* ep_scale; const float* ep_shift; const float* ep_res; int ep_act;
  float drop_scale; uint32_t drop_thresh, drop_key;
  int prio;                        // 1: issue priority rises through a chunk's MFMA steps (4-wave kernel; see its main loop)
  WINO_DIAG_FIELDS                 // empty in the shipped build (conv_wino_diag.h: diagnostic builds -DFS_WINO_TRACE / -DFS_WINO_CLOCK)
};

// Weight pack: Up[g4 = ((chunk*3 + ky)*4 + c)*2 + s][plane][n][j] = plane-th term of U_c of filter row ky at
// (k = 32*chunk + 16*s + j, n), scaled by 2^(14-Ew) in f16x2 (|U| <= 1.5 max|w| stays inside fp16), behind a HDR-byte header.
//   forward : row ky = W[ky*3 + 0..2][k][n]                     (K = Cin,  N = Cout)
//   bwd-data: row ky = W[8 - (ky*3 + 0..2)][n][k]               (K = Cout, N = Cin; taps flipped)
template <class P>
__global__ __launch_bounds__(256) void wino_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew,
                                                        int Cin, int Cout, int transposed, int Ks, int Ns, int Npad, long total) {
  // one thread per (chunk, ky, 8-k quarter, n): the 3 taps of 8 k are loaded once (24 independent loads, clamped addresses)
  // and all 4 components are formed from them
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int qk = (int)(idx / Npad);
  const int quarter = qk & 3, kyc = qk >> 2;                   // quarter = s * 2 + half of the 16-k fragment
  const int chunk = kyc / 3, ky = kyc - 3 * chunk;
  const int k0 = chunk * 32 + quarter * 8;
  const bool nok = n < Ns;
  const int nc = nok ? n : 0;
  float g[3][8];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int tap = ky * 3 + kx;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = k0 + j;
      const bool ok = nok && k < Ks;
      const int kc = ok ? k : 0;
      const float v = transposed ? w[((long)(8 - tap) * Cin + nc) * Cout + kc] : w[((long)tap * Cin + kc) * Cout + nc];
      g[kx][j] = ok ? v * sc : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    typename P::x8 p[P::NPL];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float u = c == 0 ? g[0][j] : (c == 3 ? g[2][j] : (c == 1 ? 0.5f * ((g[0][j] + g[2][j]) + g[1][j]) : 0.5f * ((g[0][j] + g[2][j]) - g[1][j])));
      typename P::T t[P::NPL];
      P::split(u, t);
#pragma unroll
      for (int pl = 0; pl < P::NPL; ++pl) p[pl][j] = t[pl];
    }
    const int g4 = (kyc * 4 + c) * 2 + (quarter >> 1);
#pragma unroll
    for (int pl = 0; pl < P::NPL; ++pl)
      *reinterpret_cast<typename P::x8*>(wp + (((long)g4 * P::NPL + pl) * Npad + n) * 16 + 8 * (quarter & 1)) = p[pl];
  }
}

constexpr int XCH_BYTES = 65536;   // epilogue exchange: [4 waves][2 parities][2 blocks][32 rows][32 columns] floats

template <class P>
constexpr int wino_lds_bytes() {
  constexpr int img = P::NPL * PLANE * 2;
  return (img > XCH_BYTES ? img : XCH_BYTES) + 4 * 32 * 2 * 4 /* stats */ + 2 * 64 * 4 /* rowpix x2 */ + 16 /* amax cells */;
}

template <class P>
__global__ __launch_bounds__(256, 2) void conv3x3_wino_kernel(WinoArgs a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  constexpr int MI = 2;                  // 32-pair blocks per wave
  constexpr int NITEM = 3;               // (slot, channel quad) items per thread: 8 * slots <= 256 * NITEM
  constexpr int IMG = NPL * PLANE * 2 > XCH_BYTES ? NPL * PLANE * 2 : XCH_BYTES;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename P::T* Ah = reinterpret_cast<typename P::T*>(smem);                  // [NPL][4][WNS][XLD]
  float* red = reinterpret_cast<float*>(smem + IMG);                           // [4 waves][32 columns][2]
  int* rowpix = reinterpret_cast<int*>(smem + IMG + 1024);                     // [2 tile parities][64]
  unsigned* amax_cell = reinterpret_cast<unsigned*>(rowpix + 128);             // [2]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);      // scalar: cp selects the wave's B stream through the SGPR offset operand
  const int cp = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  WINO_DIAG_KERNEL_BEGIN;
  WINO_STAMP(0);
  // ---- persistent schedule: XCD x owns a contiguous range of tiles (the column tiles of one pixel tile and neighbouring halos meet
  // in one 4 MB L2); its workgroups (blockIdx & 7 == x under round-robin placement -- speed only) take every L-th tile of it ----
  const int ntile = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = ntile >> 3, rm = ntile & 7;
  const int t_first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd);
  const int t_end = t_first + qd + (xcd < rm ? 1 : 0);
  const int L = ((int)gridDim.x + 7 - xcd) >> 3;
  int wg = t_first + loc;
  if (wg >= t_end) return;               // workgroup-uniform

  const int nslots = (a.Ph + 2) * a.PP, npairs = a.Ph * a.PP;
  // (image, row) of virtual row vy; gap rows and rows outside the stacked batch get row = H (invalid)
  auto image_row = [&](int vy, int& bb, int& yy) {
    const bool in = vy >= 0 && vy < a.B * a.Hv;
    bb = in ? div_small(vy, a.magic_hv) : 0;
    yy = in ? vy - bb * a.Hv : a.H;
  };
  // tile index -> pixel tile mt, first column n0, tile origin (y0 in stacked rows, x0)
  auto decode = [&](int w_, int& mt, int& n0, int& y0, int& x0) {
    mt = div_small1(w_, a.magic_ny);
    n0 = (w_ - mt * a.ny) * 64;
    const int ty = div_small1(mt, a.magic_tx), tx = mt - ty * a.tiles_x;
    y0 = ty * a.Ph; x0 = tx * 2 * a.PP;
  };
  // (the per-lane constants of the two set-up lambdas are recomputed per tile from an opaque copy of tid: hoisted out of the tile loop
  // they would sit in -- and spill from -- registers the MFMA loop needs)
  auto write_rowpix = [&](int par, int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < 32 * MI) {
      const int tid = t_;
      const int p = (tid & ~31) + row_perm(tid & 31);
      const int py = div_small(p, a.magic_pp), px = p - py * a.PP;
      int bb, yy;
      image_row(y0 + py, bb, yy);
      const bool live = p < npairs && yy < a.H && x0 + 2 * px < a.W;
      rowpix[par * 64 + tid] = live ? ((bb * a.H + yy) * a.W + x0 + 2 * px) * a.Cd : -1;      // element offset of the pair's even pixel
    }
  };
  const int q = tid & 7;
  int goff[NITEM], gmask = 0;            // gmask: 4 validity bits per item
  auto setup_loader = [&](int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (t_ >> 3) + 32 * i;
      const int hy = div_small(slot, a.magic_pp), pj = slot - hy * a.PP;
      const int ix = x0 + 2 * pj - 1;
      int bb, iy;
      image_row(y0 + hy - 1, bb, iy);
      const bool rowok = slot < nslots && iy < a.H;
      int m = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) m |= (rowok && ix + e >= 0 && ix + e < a.W) ? (1 << e) : 0;
      gmask = i == 0 ? m : (gmask | (m << (4 * i)));
      goff[i] = ((bb * a.H + (rowok ? iy : 0)) * a.W + ix) * a.Cs + 4 * (t_ & 7);
    }
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);

  f32x4 ra[NITEM][4];
  auto load_halo = [&](int chunk) {
    const int c0 = chunk * 32;
    const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int i = 0; i < NITEM; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = cok && ((gmask >> (4 * i + e)) & 1);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + e * a.Cs + c0) * 4u) : (int)OOB, 0, 0);
        ra[i][e] = __builtin_bit_cast(f32x4, v);
      }
  };
  auto transform = [&]() {               // d0..d3 -> T0..T3 in place
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const f32x4 d0 = ra[i][0], d1 = ra[i][1], d2 = ra[i][2], d3 = ra[i][3];
      ra[i][0] = d0 - d2; ra[i][1] = d1 + d2; ra[i][2] = d2 - d1; ra[i][3] = d1 - d3;
    }
  };
  auto tile_amax = [&](int cell) {       // max |T| of the transformed registers -> LDS cell
    float m = 0.f;
#pragma unroll
    for (int i = 0; i < NITEM; ++i)
#pragma unroll
      for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ra[i][c][e]));
    m = wave_max(m);
    if (lane == 0) atomicMax(&amax_cell[cell], __builtin_bit_cast(unsigned, m));
  };
  auto store_halo = [&](float sc) {
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (tid >> 3) + 32 * i;
      if (slot >= nslots) continue;
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X4 p[NPL];
        P::split4(P::SCALED ? ra[i][c] * sc : ra[i][c], p);
        const int o = (c * WNS + slot) * XLD + 4 * q;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ah[pl * PLANE + o]) = p[pl];
      }
    }
  };

  int rowbase[MI];                       // [mi]: element offset of the wave's first component image, filter row 0 (tile-invariant)
  const int rowstep = a.PP * XLD;        // elements per halo row (scalar)
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = 32 * mi + row_perm(l31);
    const bool live = p < npairs;
    const int py = live ? div_small(p, a.magic_pp) : 0, px = live ? p - py * a.PP : 0;
    rowbase[mi] = 2 * cp * CPLANE + (py * a.PP + px) * XLD + 8 * lh;
  }
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = a.nchunk * 12;           // B fragments this wave consumes per tile
  auto b_voff = [&](int n0) { return HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2; };

  X8 fa[2][MI][NPL];      // [buffer][mi][plane]
  constexpr int RD = 3;   // B-fragment ring: fragments run two steps ahead of the MFMAs (deeper rings measured +-0.5 %, profiles/r02)
  X8 fb[RD][NPL];         // [ring slot][plane]
  auto load_b = [&](int gg, int voff, X8 (&dst)[NPL]) {
    const int g4 = (gg >> 2) * 8 + 4 * cp + (gg & 3);      // wave-uniform: goes in the scalar offset operand
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, g4 * step_bytes + pl * plane_bytes, 0);
      dst[pl] = __builtin_bit_cast(X8, v);
    }
  };
  auto read_a = [&](int step, X8 (&dst)[MI][NPL]) {
    const int ky = step >> 2, ci = (step >> 1) & 1, s2 = step & 1;
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        dst[mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase[mi] + ky * rowstep + ci * CPLANE + 16 * s2]);
  };

  // ---- first tile ----
  int mt, n0, y0, x0;
  decode(wg, mt, n0, y0, x0);
  setup_loader(y0, x0);
  int par = 0;
  write_rowpix(par, y0, x0);
  if (tid < 2) amax_cell[tid] = 0u;
  int bvoff = b_voff(n0);
#pragma unroll
  for (int r = 0; r < RD - 1; ++r) load_b(r < G ? r : G - 1, bvoff, fb[r]);
  load_halo(0);
  __syncthreads();                        // amax cells zeroed before the first atomic
  WINO_STAMP(1);
  int cc = 0;                             // chunks processed by this workgroup (parity of the f16x2 amax cell)

  for (;;) {
    f32x16 acc[2][2];       // [component of the pair][pair block mi]
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ci][b][r] = 0.f;
    int E = EMIN;
    int g = 0;
    const int wg_next = wg + L;
    const bool has_next = wg_next < t_end;
    int mt_n = 0, n0_n = 0, y0_n = 0, x0_n = 0, bvoff_n = bvoff;
    if (has_next) {
      decode(wg_next, mt_n, n0_n, y0_n, x0_n);
      bvoff_n = b_voff(n0_n);
    }
    for (int chunk = 0; chunk < a.nchunk; ++chunk, ++cc) {
      transform();
      if (P::SCALED) tile_amax(cc & 1);
      if (chunk < 4) WINO_STAMP(2 + 5 * chunk);
      __syncthreads();                      // amax complete; every wave has finished reading the previous image / exchange buffer
      if (chunk < 4) WINO_STAMP(3 + 5 * chunk);
      if (P::SCALED) {
        const int ec = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[cc & 1]));
        if (ec > E) {
          if (chunk > 0) {
            const float f = pow2f(E - ec);
#pragma unroll
            for (int ci = 0; ci < 2; ++ci)
#pragma unroll
              for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[ci][b][r] *= f;
          }
          E = ec;
        }
        if (tid == 0) amax_cell[(cc + 1) & 1] = 0u;
      }
      store_halo(pow2f(14 - E));
      if (chunk < 4) WINO_STAMP(4 + 5 * chunk);
      __syncthreads();
      if (chunk < 4) WINO_STAMP(5 + 5 * chunk);
      read_a(0, fa[0]);
#pragma unroll
      for (int step = 0; step < 12; ++step) {
        // a.prio: the SIMD arbitrates its two resident waves by priority, then age -- at equal priority the workgroup placed first takes the
        // matrix pipe whenever it wants it (conv_wgrad.hip, FS_WGRAD_TRACE); a priority that grows through the chunk lets the wave that
        // entered its MFMA steps first finish them, and the two workgroups of the CU alternate
        if (a.prio && (step & 3) == 0) {
          if (step == 0) __builtin_amdgcn_s_setprio(1);
          else if (step == 4) __builtin_amdgcn_s_setprio(2);
          else __builtin_amdgcn_s_setprio(3);
        }
        if (step + 1 < 12) read_a(step + 1, fa[(step + 1) & 1]);
        if (step == 0 && chunk + 1 < a.nchunk) load_halo(chunk + 1);     // (at step 10, behind the chunk's last own fragment: +-0 on the step)
        {
          const int gi = g + RD - 1;         // past this tile's last fragment: the next tile's first ones (or a repeat of the last)
          const bool own = gi < G;
          const int gn = gi - G < G ? gi - G : G - 1;
          load_b(own ? gi : (has_next ? gn : G - 1), own ? bvoff : bvoff_n, fb[(step + RD - 1) % RD]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const X8(&A)[MI][NPL] = fa[step & 1];
        const X8(&Bf)[NPL] = fb[step % RD];
        const int ci = (step >> 1) & 1;
#pragma unroll
        for (int t = 0; t < P::NTERM; ++t) {         // smallest terms first, the two blocks interleaved
          acc[ci][0] = P::mfma(A[0][P::ta(t)], Bf[P::tb(t)], acc[ci][0]);
          acc[ci][1] = P::mfma(A[1][P::ta(t)], Bf[P::tb(t)], acc[ci][1]);
        }
        __builtin_amdgcn_sched_barrier(0);
        ++g;
      }
      if (a.prio) __builtin_amdgcn_s_setprio(0);
      if (chunk < 4) WINO_STAMP(6 + 5 * chunk);
    }

    // ---- epilogue: inverse transform across the two component-pair waves through LDS, rows of [pixel][4 channels] out ----
    WINO_STAMP(22);
    __syncthreads();                        // the halo image is dead
    WINO_STAMP(23);
    {
      // partial sums in accumulator order: parity 0 (even pixel) = m0 + m1 | m2, parity 1 (odd pixel) = m1 | -(m2 + m3)
      int l_ = lane;
      asm volatile("" : "+v"(l_));         // (epilogue lane constants are formed per tile, not kept across the MFMA loop)
      float* xw = reinterpret_cast<float*>(smem) + wave * 4096 + (4 * (l_ >> 5)) * 32 + (l_ & 31);      // [parity][block][32 rows][32 columns]
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          const float ev = cp == 0 ? acc[0][b][r] + acc[1][b][r] : acc[0][b][r];
          const float od = cp == 0 ? acc[1][b][r] : -(acc[0][b][r] + acc[1][b][r]);
          xw[(b * 32 + row) * 32] = ev;
          xw[2048 + (b * 32 + row) * 32] = od;
        }
    }
    // bwd-data extras (BatchNorm-backward sums, residual addend): a four-row window of their operands is kept in flight. Rows 0-3 are
    // requested HERE, before the exchange barrier, and row k + 4 when row k has been consumed, so the HBM latency overlaps the barrier,
    // the LDS reads and four rows of stores instead of sitting in front of every row
    f32x4 pf_y[4], pf_a[4];
    unsigned pf_ym[4], pf_am[4];
    const bool want_y = a.bn_y != nullptr, want_a = a.add_src != nullptr;
    // (issued without a branch: a conditional definition would make the window loop-carried and spill it over the MFMA loop; an absent
    // operand has a zero-size descriptor, whose loads return 0 without touching memory)
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    auto pf_issue = [&](int slot, int i) {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int n_ = n0 + 32 * wn + (l_ & 7) * 4;
      const int pix = rowpix[par * 64 + (i >> 2) * 32 + 8 * (i & 3) + (l_ >> 3)];
      const bool live = pix >= 0 && n_ < a.Cd;
      const unsigned e = (unsigned)(pix + n_ + cp * a.Cd);
      const int o16 = live ? (int)(e * 4u) : (int)OOB, o1 = live ? (int)(e >> 2) : (int)OOB;
      pf_y[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, o16, 0, 0));
      pf_ym[slot] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_ym, o1, 0, 0) | ym_all;
      pf_a[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, o16, 0, 0));
      pf_am[slot] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_am, o1, 0, 0) | am_all;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) pf_issue(i, i);
    __syncthreads();
    if (has_next) {          // the accumulators are dead: the next tile's first halo image travels while this tile's rows are stored
      setup_loader(y0_n, x0_n);
      write_rowpix(par ^ 1, y0_n, x0_n);
      load_halo(0);
    }
    float f1 = 1.f, f2 = 1.f;
    if (P::SCALED) {
      const int Ew = exponent_of_bits(*a.ew);
      const int es = E + Ew - 28;
      const bool one = es >= -126 && es <= 127;
      f1 = one ? pow2f(es) : pow2f(E - 14);
      f2 = one ? 1.f : pow2f(Ew - 14);
    }
    {
      // this wave stores pixel parity cp of its 32-column half: lane = (row within a group of 8, channel quad)
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int c4 = (l_ & 7) * 4, rsub = l_ >> 3;
      const int n = n0 + 32 * wn + c4;
      const bool nok = n < a.Cd;
      f32x4 bv = {0.f, 0.f, 0.f, 0.f};
      if (a.bias != nullptr && nok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) bv[j] = a.bias[n + j];
      }
      f32x4 ep_sc = {1.f, 1.f, 1.f, 1.f}, ep_sh = {0.f, 0.f, 0.f, 0.f};
      if (a.ep_scale != nullptr && nok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { ep_sc[j] = a.ep_scale[n + j]; ep_sh[j] = a.ep_shift[n + j]; }
      }
      f32x4 bn_mu = {0.f, 0.f, 0.f, 0.f}, bn_is = {0.f, 0.f, 0.f, 0.f};
      if (a.bn_y != nullptr && nok) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { bn_mu[j] = a.bn_mean[n + j]; bn_is[j] = a.bn_invstd[n + j]; }
      }
      const float* mine = reinterpret_cast<const float*>(smem) + wave * 4096 + cp * 2048;
      const float* theirs = reinterpret_cast<const float*>(smem) + (wave ^ 2) * 4096 + cp * 2048;
      f32x4 csum = {0.f, 0.f, 0.f, 0.f}, csq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = b * 32 + 8 * k + rsub;
          const int pix = rowpix[par * 64 + row];
          const f32x4 m = *reinterpret_cast<const f32x4*>(mine + row * 32 + c4) + *reinterpret_cast<const f32x4*>(theirs + row * 32 + c4);
          const bool live = pix >= 0 && nok;
          const unsigned e = (unsigned)(pix + n + cp * a.Cd);                 // the {2,3} waves write the odd pixel of the pair
          f32x4 v;
          const uint32_t keep = a.drop_thresh != 0u ? fs_dropout_keep4((uint32_t)e, a.drop_key, a.drop_thresh) : 15u;      // e is a multiple of 4
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = P::SCALED ? fmaf(m[j] * f2, f1, bv[j]) : m[j] + bv[j];
            if (a.drop_thresh != 0u) x = ((keep >> j) & 1u) ? x * a.drop_scale : 0.f;
            v[j] = live ? x : 0.f;
          }
          if (a.ep_scale != nullptr && live) {
            v = v * ep_sc + ep_sh;
            if (a.ep_res != nullptr) v += *reinterpret_cast<const f32x4*>(a.ep_res + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fs_act(v[j], a.ep_act);
          }
          if (want_a && live) {
            f32x4 r = pf_a[k];
            const unsigned mk = pf_am[k];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = ((mk >> j) & 1u) ? r[j] : 0.f;
            v += r;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
          if (want_y) {                      // BatchNorm-backward sums of the layer whose output gradient this is
            if (live) {
              const f32x4 yy = pf_y[k];
              const unsigned mk = pf_ym[k];
              f32x4 gq = v;
#pragma unroll
              for (int j = 0; j < 4; ++j) gq[j] = ((mk >> j) & 1u) ? gq[j] : 0.f;
              csum += gq; csq += gq * ((yy - bn_mu) * bn_is);
            }
          } else {
            csum += v; csq += v * v;
          }
          if (b == 0) pf_issue(k, 4 + k);
        }
      if (a.stats != nullptr) {
        // column sums over the wave's 64 rows: lanes with equal channel quad (lane & 7) hold different rows
#pragma unroll
        for (int o = 8; o < 64; o <<= 1)
#pragma unroll
          for (int j = 0; j < 4; ++j) { csum[j] += __shfl_xor(csum[j], o, 64); csq[j] += __shfl_xor(csq[j], o, 64); }
        if (l_ < 8) {
#pragma unroll
          for (int j = 0; j < 4; ++j) { red[(wave * 32 + c4 + j) * 2] = csum[j]; red[(wave * 32 + c4 + j) * 2 + 1] = csq[j]; }
        }
        __syncthreads();
        if (tid < 128) {
          const int col = tid >> 1, which = tid & 1;
          const int w0 = col >> 5, c31 = col & 31;                          // waves w0 (even pixels) and w0 + 2 (odd pixels)
          const float v = red[(w0 * 32 + c31) * 2 + which] + red[((w0 + 2) * 32 + c31) * 2 + which];
          if (n0 + col < a.Cd) a.stats[((long)mt * a.Cd + n0 + col) * 2 + which] = v;
        }
      }
    }
    WINO_STAMP(24);
    if (!has_next) break;
    WINO_DIAG_NEXT_TILE;
    wg = wg_next; mt = mt_n; n0 = n0_n; bvoff = bvoff_n; par ^= 1;
  }
  WINO_DIAG_KERNEL_END;
}

// ---- eight-wave variant with the split work INSIDE the MFMA phase (bf16x3, round 3) ----------------------------------------------
// What the phase stamps of the kernel above say (tools/wino_trace.sh, profiles/r03/wino_phase_stamps.txt): its two co-resident
// workgroups run in lockstep -- both split and store a halo image at the same time (about 3 k cycles without one MFMA on the CU),
// then both multiply at the same time (2 x 144 MFMAs in 9.5 k cycles: the pipe is ~95 % busy in THAT phase).  Holding the two
// workgroups one phase apart (a ping-pong of two 4-wave halves behind one barrier, measured and rejected) does not help: a wave
// multiplying alone needs 7.6 k cycles for its 144 MFMAs, because every vector-memory instruction in its stream (36 weight-fragment
// loads + 12 halo loads per chunk) costs ~60 issue cycles the pipe sits idle unless a partner wave multiplies meanwhile.
// So here both waves of a SIMD stay in the MFMA phase all the time and the split/store work of the NEXT chunk is a block of VALU
// code inside the CURRENT chunk's MFMA phase, placed at a different step for waves 0-3 than for waves 4-7 (SIMD partners are waves
// w and w + 4): while one wave of a SIMD splits, its partner multiplies.  That needs the halo image double-buffered in LDS
// (2 x 76.8 KB), i.e. ONE workgroup of eight waves per CU sharing it: 64 pairs x 128 columns, waves = 2 component pairs x 4 column
// groups, so the split work per MFMA is halved as well.
// (A 64-column form -- 2 x 2 waves x the two 16-channel halves of every chunk, added in the epilogue -- was 19 % slower than the
// 4-wave kernel on 64 -> 64 @ 80x80 and is not kept; which layers come here: wino_use8 below.)
// One barrier per chunk.  Halo loads for chunk j + 2 are issued right behind the split of chunk j + 1 into the registers it freed
// (a whole chunk period of cover).  Tile loop persistent as above; BatchNorm partial sums: one slab row per (pixel tile, pixel
// parity) -- no exchange between waves.
template <class P>
__global__ __launch_bounds__(512, 1) void conv3x3_wino8_kernel(WinoArgs a) {
  static_assert(!P::SCALED, "no slot for the f16x2 tile-maximum exchange in this schedule");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  constexpr int MI = 2;                    // 32-pair blocks per wave
  constexpr int NITEM = 2;                 // (slot, channel quad) items per thread: 8 * slots <= 512 * NITEM
  constexpr int IMG = NPL * PLANE * 2;     // bytes of one halo image
  constexpr int NCOL = 128;                // columns per workgroup
  constexpr int NSTEP = 12;                // MFMA steps of a chunk
  constexpr int S_EARLY = 2, S_LATE = 8;   // steps behind which waves 0-3 / 4-7 split the next chunk
  static_assert(IMG >= XCH_BYTES, "an exchange round (one pair block of all 8 waves) lives in one image buffer");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* rowpix = reinterpret_cast<int*>(smem + 2 * IMG);                        // [2 tile parities][64]

  const int tid = threadIdx.x, lane = tid & 63;
  const int hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cp = hw >> 2;                                                      // component pair {0,1} / {2,3}; also the stagger group
  const int wn = hw & 3;                                                       // 32-column group
  const int l31 = lane & 31, lh = lane >> 5;

  // ---- persistent schedule (as above, one workgroup per CU) ----
  const int ntile = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = ntile >> 3, rm = ntile & 7;
  const int t_first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd);
  const int t_end = t_first + qd + (xcd < rm ? 1 : 0);
  const int L = ((int)gridDim.x + 7 - xcd) >> 3;
  if (t_first + loc >= t_end) return;                                          // workgroup-uniform
  const int ntw = (t_end - t_first - loc + L - 1) / L;                         // tiles this workgroup walks: t_first + loc + k L
  const int nch = a.nchunk;
  const int nslots = (a.Ph + 2) * a.PP, npairs = a.Ph * a.PP;

  auto image_row = [&](int vy, int& bb, int& yy) {
    const bool in = vy >= 0 && vy < a.B * a.Hv;
    bb = in ? div_small(vy, a.magic_hv) : 0;
    yy = in ? vy - bb * a.Hv : a.H;
  };
  auto decode = [&](int w_, int& mt, int& n0, int& y0, int& x0) {
    mt = div_small1(w_, a.magic_ny);
    n0 = (w_ - mt * a.ny) * NCOL;
    const int ty = div_small1(mt, a.magic_tx), tx = mt - ty * a.tiles_x;
    y0 = ty * a.Ph; x0 = tx * 2 * a.PP;
  };
  auto write_rowpix = [&](int par, int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < 32 * MI) {
      const int p = (t_ & ~31) + row_perm(t_ & 31);
      const int py = div_small(p, a.magic_pp), px = p - py * a.PP;
      int bb, yy;
      image_row(y0 + py, bb, yy);
      const bool live = p < npairs && yy < a.H && x0 + 2 * px < a.W;
      rowpix[par * 64 + t_] = live ? ((bb * a.H + yy) * a.W + x0 + 2 * px) * a.Cd : -1;
    }
  };
  // halo loader: item i of a thread = (slot (tid >> 3) + 64 i, channel quad tid & 7); item 1 exists for slots 64..79 (waves 0, 1)
  int goff[NITEM], gmask = 0;
  auto setup_loader = [&](int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (t_ >> 3) + 64 * i;
      const int hy = div_small(slot, a.magic_pp), pj = slot - hy * a.PP;
      const int ix = x0 + 2 * pj - 1;
      int bb, iy;
      image_row(y0 + hy - 1, bb, iy);
      const bool rowok = slot < nslots && iy < a.H;
      int m = 0;
#pragma unroll
      for (int e = 0; e < 4; ++e) m |= (rowok && ix + e >= 0 && ix + e < a.W) ? (1 << e) : 0;
      gmask = i == 0 ? m : (gmask | (m << (4 * i)));
      goff[i] = ((bb * a.H + (rowok ? iy : 0)) * a.W + ix) * a.Cs + 4 * (t_ & 7);
    }
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
  const int q = tid & 7;
  const bool item1 = (tid >> 3) + 64 < nslots;                                 // wave-uniform in groups of 8 lanes; false for waves >= 2

  f32x4 ra[NITEM][4];
  auto load_halo = [&](int chunk) {
    const int c0 = chunk * 32;
    const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      if (i == 1 && hw >= 2) continue;                                         // waves 2..7 own no second item (wave-uniform)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = cok && ((gmask >> (4 * i + e)) & 1);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + e * a.Cs + c0) * 4u) : (int)OOB, 0, 0);
        ra[i][e] = __builtin_bit_cast(f32x4, v);
      }
    }
  };
  auto store_halo = [&](int buf) {         // d0..d3 -> T0..T3 -> planes -> image `buf`
    typename P::T* Aw = reinterpret_cast<typename P::T*>(smem + buf * IMG);
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      if (i == 1 && hw >= 2) continue;
      const int slot = (tid >> 3) + 64 * i;
      if (slot >= nslots) continue;
      const f32x4 d0 = ra[i][0], d1 = ra[i][1], d2 = ra[i][2], d3 = ra[i][3];
      const f32x4 T[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X4 p[NPL];
        P::split4(T[c], p);
        const int o = (c * WNS + slot) * XLD + 4 * q;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Aw[pl * PLANE + o]) = p[pl];
      }
    }
  };
  (void)item1;

  int rowbase[MI];
  const int rowstep = a.PP * XLD;
#pragma unroll
  for (int mi = 0; mi < MI; ++mi) {
    const int p = 32 * mi + row_perm(l31);
    const bool live = p < npairs;
    const int py = live ? div_small(p, a.magic_pp) : 0, px = live ? p - py * a.PP : 0;
    rowbase[mi] = 2 * cp * CPLANE + (py * a.PP + px) * XLD + 8 * lh;
  }
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = nch * NSTEP;               // B fragments this wave consumes per tile
  auto b_voff = [&](int n0) { return HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2; };

  X8 fa[2][MI][NPL];
  constexpr int RD = 3;
  X8 fb[RD][NPL];
  auto load_b = [&](int gg, int voff, X8 (&dst)[NPL]) {
    const int g4 = (gg >> 2) * 8 + 4 * cp + (gg & 3);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, g4 * step_bytes + pl * plane_bytes, 0);
      dst[pl] = __builtin_bit_cast(X8, v);
    }
  };
  auto read_a = [&](int buf, int step, X8 (&dst)[MI][NPL]) {
    const int ky = step >> 2, ci = (step >> 1) & 1, s2 = step & 1;
    const typename P::T* Ar = reinterpret_cast<const typename P::T*>(smem + buf * IMG);
#pragma unroll
    for (int mi = 0; mi < MI; ++mi)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl)
        dst[mi][pl] = *reinterpret_cast<const X8*>(&Ar[pl * PLANE + rowbase[mi] + ky * rowstep + ci * CPLANE + 16 * s2]);
  };

  // ---- cursors: the halo loader runs up to two chunks ahead of the MFMAs, across tile boundaries ----
  int lk = 0, lc = 0;                      // (tile number, chunk) of the next halo load
  auto loader_advance = [&]() {            // after a load: move on; entering a new tile re-aims the loader
    if (++lc == nch) {
      lc = 0; ++lk;
      if (lk < ntw) {
        int mt_, n0_, y0_, x0_;
        decode(t_first + loc + lk * L, mt_, n0_, y0_, x0_);
        setup_loader(y0_, x0_);
      }
    }
  };
  int mt, n0, y0, x0;
  decode(t_first + loc, mt, n0, y0, x0);
  setup_loader(y0, x0);
  write_rowpix(0, y0, x0);
  int bvoff = b_voff(n0);
#pragma unroll
  for (int r = 0; r < RD - 1; ++r) load_b(r, bvoff, fb[r]);
  load_halo(0);
  loader_advance();
  store_halo(0);                           // chunk 0 of the first tile (exposed once per workgroup)
  if (lk < ntw) { load_halo(lc); loader_advance(); }
  __syncthreads();

  int J = 0;                               // chunks multiplied so far: image buffer of the current chunk = J & 1
  const int total_chunks = ntw * nch;
  for (int k = 0; k < ntw; ++k) {
    const bool has_next = k + 1 < ntw;
    int mt_n = 0, n0_n = 0, y0_n = 0, x0_n = 0, bvoff_n = bvoff;
    if (has_next) {
      decode(t_first + loc + (k + 1) * L, mt_n, n0_n, y0_n, x0_n);
      bvoff_n = b_voff(n0_n);
    }
    f32x16 acc[2][2];
#pragma unroll
    for (int ci = 0; ci < 2; ++ci)
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[ci][b][r] = 0.f;
    int g = 0;
    for (int chunk = 0; chunk < nch; ++chunk, ++J) {
      const int buf = J & 1;
      const bool split_next = J + 1 < total_chunks;          // ra holds the halo of chunk J + 1 (this tile's next chunk or the next tile's first)
      read_a(buf, 0, fa[0]);
#pragma unroll
      for (int st = 0; st < NSTEP; ++st) {
        if (st + 1 < NSTEP) read_a(buf, st + 1, fa[(st + 1) & 1]);
        {
          const int gi = g + RD - 1;
          const bool own = gi < G;
          const int gn = gi - G < G ? gi - G : G - 1;
          load_b(own ? gi : (has_next ? gn : G - 1), own ? bvoff : bvoff_n, fb[(st + RD - 1) % RD]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const X8(&A)[MI][NPL] = fa[st & 1];
        const X8(&Bf)[NPL] = fb[st % RD];
        const int ci = (st >> 1) & 1;
        if (ci == 0) {
#pragma unroll
          for (int t = 0; t < P::NTERM; ++t) {
            acc[0][0] = P::mfma(A[0][P::ta(t)], Bf[P::tb(t)], acc[0][0]);
            acc[0][1] = P::mfma(A[1][P::ta(t)], Bf[P::tb(t)], acc[0][1]);
          }
        } else {
#pragma unroll
          for (int t = 0; t < P::NTERM; ++t) {
            acc[1][0] = P::mfma(A[0][P::ta(t)], Bf[P::tb(t)], acc[1][0]);
            acc[1][1] = P::mfma(A[1][P::ta(t)], Bf[P::tb(t)], acc[1][1]);
          }
        }
        __builtin_amdgcn_sched_barrier(0);
        ++g;
        // the NEXT chunk's split / store, at a different step for the two waves of a SIMD (w and w + 4, i.e. cp = 0 / 1)
        if ((st == S_EARLY || st == S_LATE) && split_next && (st == S_EARLY) == (cp == 0)) {
          store_halo(buf ^ 1);
          if (lk < ntw) { load_halo(lc); loader_advance(); }    // chunk J + 2 into the registers just freed
        }
      }
      __syncthreads();
    }

    // ---- epilogue: two exchange rounds (pair block 0, then 1) through the image buffer the last chunk was read from ----
    unsigned char* const xbuf = smem + ((J - 1) & 1) * IMG;
    const int par = k & 1;
    const int n = n0 + 32 * wn + (lane & 7) * 4;
    const bool nok = n < a.Cd;
    f32x4 bv = {0.f, 0.f, 0.f, 0.f};
    if (a.bias != nullptr && nok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) bv[j] = a.bias[n + j];
    }
    f32x4 csum = {0.f, 0.f, 0.f, 0.f}, csq = {0.f, 0.f, 0.f, 0.f};
    const bool want_y = a.bn_y != nullptr, want_a = a.add_src != nullptr;
    // four-row window of the bwd-data extras' operands (BatchNorm-backward sums, residual addend), as in the 4-wave kernel: round 0's rows
    // are requested before the first exchange barrier, round 1's row k when round 0's row k has been consumed; issued without a branch
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    f32x4 pf_y[4], pf_a[4];
    unsigned pf_ym[4], pf_am[4];
    auto pf_issue = [&](int slot, int i) {
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int pix = rowpix[par * 64 + (i >> 2) * 32 + 8 * (i & 3) + (l_ >> 3)];
      const bool live = pix >= 0 && nok;
      const unsigned e = (unsigned)(pix + n + cp * a.Cd);
      const int o16 = live ? (int)(e * 4u) : (int)OOB, o1 = live ? (int)(e >> 2) : (int)OOB;
      pf_y[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_y, o16, 0, 0));
      pf_ym[slot] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_ym, o1, 0, 0) | ym_all;
      pf_a[slot] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rs_a, o16, 0, 0));
      pf_am[slot] = (unsigned)__builtin_amdgcn_raw_buffer_load_b8(rs_am, o1, 0, 0) | am_all;
    };
#pragma unroll
    for (int i = 0; i < 4; ++i) pf_issue(i, i);
    f32x4 ep_sc = {1.f, 1.f, 1.f, 1.f}, ep_sh = {0.f, 0.f, 0.f, 0.f};
    if (a.ep_scale != nullptr && nok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { ep_sc[j] = a.ep_scale[n + j]; ep_sh[j] = a.ep_shift[n + j]; }
    }
    f32x4 bn_mu = {0.f, 0.f, 0.f, 0.f}, bn_is = {0.f, 0.f, 0.f, 0.f};
    if (a.bn_y != nullptr && nok) {
#pragma unroll
      for (int j = 0; j < 4; ++j) { bn_mu[j] = a.bn_mean[n + j]; bn_is[j] = a.bn_invstd[n + j]; }
    }
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      {
        // partial sums in accumulator order: parity 0 (even pixel) = m0 + m1 | m2, parity 1 (odd pixel) = m1 | -(m2 + m3)
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        float* xw = reinterpret_cast<float*>(xbuf) + hw * 2048 + (4 * (l_ >> 5)) * 32 + (l_ & 31);      // [wave][parity][32 rows][32 columns]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          const float ev = cp == 0 ? acc[0][b][r] + acc[1][b][r] : acc[0][b][r];
          const float od = cp == 0 ? acc[1][b][r] : -(acc[0][b][r] + acc[1][b][r]);
          xw[row * 32] = ev;
          xw[1024 + row * 32] = od;
        }
      }
      __syncthreads();
      // this wave stores pixel parity cp of its column group
      {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int c4 = (l_ & 7) * 4, rsub = l_ >> 3;
        // source waves: both component pairs (hw and hw ^ 4)
        const float* s0 = reinterpret_cast<const float*>(xbuf) + hw * 2048 + cp * 1024;
        const float* s1 = reinterpret_cast<const float*>(xbuf) + (hw ^ 4) * 2048 + cp * 1024;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int row = 8 * kk + rsub;
          const int pix = rowpix[par * 64 + b * 32 + row];
          const f32x4 m = *reinterpret_cast<const f32x4*>(s0 + row * 32 + c4) + *reinterpret_cast<const f32x4*>(s1 + row * 32 + c4);
          const bool live = pix >= 0 && nok;
          const unsigned e = (unsigned)(pix + n + cp * a.Cd);                 // the {2,3} waves write the odd pixel of the pair
          f32x4 v;
          const uint32_t keep = a.drop_thresh != 0u ? fs_dropout_keep4((uint32_t)e, a.drop_key, a.drop_thresh) : 15u;      // e is a multiple of 4
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            float x = m[j] + bv[j];
            if (a.drop_thresh != 0u) x = ((keep >> j) & 1u) ? x * a.drop_scale : 0.f;
            v[j] = live ? x : 0.f;
          }
          if (a.ep_scale != nullptr && live) {
            v = v * ep_sc + ep_sh;
            if (a.ep_res != nullptr) v += *reinterpret_cast<const f32x4*>(a.ep_res + e);
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = fs_act(v[j], a.ep_act);
          }
          if (want_a && live) {
            f32x4 r = pf_a[kk];
            const unsigned mk = pf_am[kk];
#pragma unroll
            for (int j = 0; j < 4; ++j) r[j] = ((mk >> j) & 1u) ? r[j] : 0.f;
            v += r;
          }
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
          if (want_y) {                      // BatchNorm-backward sums of the layer whose output gradient this is
            if (live) {
              const f32x4 yy = pf_y[kk];
              const unsigned mk = pf_ym[kk];
              f32x4 gq = v;
#pragma unroll
              for (int j = 0; j < 4; ++j) gq[j] = ((mk >> j) & 1u) ? gq[j] : 0.f;
              csum += gq; csq += gq * ((yy - bn_mu) * bn_is);
            }
          } else {
            csum += v; csq += v * v;
          }
          if (b == 0) pf_issue(kk, 4 + kk);
        }
      }
      __syncthreads();
    }
    if (a.stats != nullptr) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { csum[j] += __shfl_xor(csum[j], o, 64); csq[j] += __shfl_xor(csq[j], o, 64); }
      if (lane < 8 && nok) {               // slab row = (pixel tile, pixel parity): [2 nx][Cd][2]
        float* dst = a.stats + ((long)(2 * mt + cp) * a.Cd + n) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[2 * j] = csum[j]; dst[2 * j + 1] = csq[j]; }
      }
    }
    if (!has_next) break;
    mt = mt_n; n0 = n0_n; bvoff = bvoff_n;
    write_rowpix((k + 1) & 1, y0_n, x0_n);
  }
}

const bool g_wino = FS_ENV_INT("FS_WINOGRAD", 1) != 0;

// Ph rows x PP pairs <= 64 pairs per workgroup, halo (Ph+2)*PP <= WNS; fewest tiles over the stacked batch, then smallest halo.
void wino_plan(int B, int H, int W, int& Ph, int& PP, int& tiles_x, int& nx) {
  const long rows = (long)B * (H + 1);
  const int wp = W / 2, maxpairs = 64, maxslots = WNS;
  long best = -1;
  Ph = 8; PP = 8;
  for (int pp = 2; pp <= 32 && pp <= wp; ++pp) {      // div_small needs a divisor >= 2
    int ph = maxpairs / pp;
    while (ph > 1 && (ph + 2) * pp > maxslots) --ph;
    if (ph < 1 || (ph + 2) * pp > maxslots) continue;
    const long tiles = (long)cdiv(rows, ph) * cdiv(wp, pp);
    const long cost = tiles * 1000 + (ph + 2) * pp;
    if (best < 0 || cost < best) { best = cost; Ph = ph; PP = pp; }
  }
  tiles_x = cdiv(wp, PP);
  nx = cdiv(rows, Ph) * tiles_x;
}

// persistent grid: two workgroups per CU of the current device (cached per device)
int wino_grid_slots() {
  static int slots[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 512;
  if (slots[dev] == 0) {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    slots[dev] = 2 * cus;
  }
  return slots[dev];
}

// Which 3x3 layers take the eight-wave kernel: bf16x3, more than 64 output channels (128-column workgroups) and at least 8 channel
// chunks -- measured in one gpurun call against the persistent 4-wave kernel: 256 -> 256 @ 20x20 139 vs 150 us, 512 -> 512 @ 10x10
// 150 vs 165 us, 960 -> 240 @ 80x80 5.81 vs 5.92 ms; 128 -> 128 @ 40x40 the same; its 64-column form (k-halves) 19 % slower on
// 64 -> 64 @ 80x80 and not kept.  FS_WINO8=0 / 2: never / every layer above 64 output channels (kernel experiments; read once).
bool wino_use8(int mode, int Cs, int Cd) {
  static const int pol = FS_ENV_INT("FS_WINO8", 1);
  if (mode != 1 || Cd <= 64 || pol == 0) return false;
  return pol == 2 || Cs >= 256;
}

template <class P>
int run_wino(WinoArgs& a, const float* w, void* ws, const unsigned* w_amax, int Cin, int Cout, int transposed, hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(w, (long)9 * Cin * Cout, ws, w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.nchunk * 12 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((wino_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws),
                       a.ew, Cin, Cout, transposed, a.Cs, a.Cd, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
  const long ntile = (long)a.nx * a.ny;
  const int slots = wino_grid_slots();
  // the dynamic-LDS opt-in (above the 64 KB default) is a per-device function attribute: set it once on every device used
  constexpr int lds = wino_lds_bytes<P>();
  constexpr int lds8 = 2 * P::NPL * PLANE * 2 + 2 * 64 * 4;
  {
    static unsigned long long done = 0ull;            // one mask per precision (template instance)
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
    if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
      hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (attr != hipSuccess) return (int)attr;
      if constexpr (!P::SCALED) {
        attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino8_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
        if (attr != hipSuccess) return (int)attr;
      }
      if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
  }
  if constexpr (!P::SCALED) {
    if (a.ny * 128 == a.Npad && wino_use8(1, a.Cs, a.Cd)) {       // planned for 128-column workgroups (fs_wino_conv3x3)
      const unsigned grid8 = (unsigned)(ntile < slots / 2 ? ntile : slots / 2);
      hipLaunchKernelGGL((conv3x3_wino8_kernel<P>), dim3(grid8), dim3(512), lds8, stream, a);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
  }
  const unsigned grid = (unsigned)(ntile < slots ? ntile : slots);
  WINO_DIAG_BEFORE_LAUNCH(a, grid, stream);
  hipLaunchKernelGGL((conv3x3_wino_kernel<P>), dim3(grid), dim3(256), lds, stream, a);
  FS_LAUNCH_CHECK();
  WINO_DIAG_AFTER_LAUNCH(a, grid, ntile, stream);
  return FS_OK;
}

}  // namespace

// f16x2 spends half the MFMAs per product, so the doubled split work of the transform only pays from 4 channel chunks up
// (profiles/r02/winograd_kernel_times.txt: 64 -> 64 @ 80x80 103 us against 94 us, 128 -> 128 @ 40x40 90.6 against 93.7)
bool fs_wino_eligible(int mode, int B, int H, int W, int Cs, int Cd) {
  if (mode == 2 && Cs < 128) return false;
  return g_wino && (mode == 1 || mode == 2) && W % 2 == 0 && W >= 4 && (long)B * (H + 1) < 65536 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 32;
}

// Which of the eligible layers take the F(4,3) kernel of conv_wino4.hip (policy from the same-box A/B of round 5, profiles/r05/wino4_ab.txt;
// FS_WINO4 in kernel A/B builds: 0 none, 1 the measured policy, 2 every eligible layer)
static bool wino4_selected(int mode, int B, int H, int W, int Cs, int Cd) {
  static const int pol = FS_ENV_INT("FS_WINO4", 1);
  if (pol == 0 || !fs_wino4_eligible(mode, B, H, W, Cs, Cd)) return false;
  // B = 64, us per call F(2,3) -> F(4,3): 64 -> 64 @ 80x80 167 -> 141 (fwd) / 165 -> 141 (bwd-data), 128 -> 128 @ 40x40 150 -> 132 / 151 -> 134,
  // 192 -> 192 @ 80x80 1 090 -> 962, 960 -> 240 @ 80x80 5 940 -> 5 285 / 7 150 -> 5 840 and 256 -> 256 @ 20x20 144 -> 130 / 145 -> 134 with the
  // eight-wave form of the F(4,3) kernel (against the eight-wave F(2,3) kernel; its four-wave form only tied there: 142 / 142)
  return true;
}

bool fs_wino_takes_f43(int mode, int B, int H, int W, int Cs, int Cd) {
  return fs_wino_eligible(mode, B, H, W, Cs, Cd) && wino4_selected(mode, B, H, W, Cs, Cd);
}

long fs_wino_pack_bytes(int mode, int Cs, int Cd) {
  const int npl = mode == 2 ? 2 : 3;
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 127) / 128) * 128;       // room for either column tiling
  const long f23 = HDR + nchunk * 24 * npl * Npad * 16 * 2;
  const long f43 = mode == 1 ? fs_wino4_pack_bytes(mode, Cs, Cd) : 0;         // (shape-independent bound: either kernel fits)
  return f23 > f43 ? f23 : f43;
}

int fs_wino_stats_slabs(int mode, int B, int H, int W, int Cs, int Cd) {
  if (wino4_selected(mode, B, H, W, Cs, Cd)) return fs_wino4_stats_slabs(B, H, W, Cs, Cd);
  int Ph, PP, tx, nx;
  wino_plan(B, H, W, Ph, PP, tx, nx);
  return wino_use8(mode, Cs, Cd) ? 2 * nx : nx;        // the eight-wave kernel writes one row per (pixel tile, pixel parity)
}

int fs_wino_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                    int B, int H, int W, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh,
                    uint32_t drop_key, const FsBnSums* bn, hipStream_t stream) {
  if (wino4_selected(mode, B, H, W, Cs, Cd))
    return fs_wino4_conv3x3(mode, src, w, bias, dst, stats, ws, B, H, W, Cs, Cd, Cin, Cout, transposed, drop_scale, drop_thresh, drop_key, bn, stream);
  WinoArgs a;
  a.bn_y = bn ? bn->y : nullptr; a.bn_mask = bn ? bn->mask : nullptr; a.bn_mean = bn ? bn->mean : nullptr; a.bn_invstd = bn ? bn->invstd : nullptr;
  a.add_src = bn ? bn->add_src : nullptr; a.add_mask = bn ? bn->add_mask : nullptr;
  a.ep_scale = bn ? bn->ep_scale : nullptr; a.ep_shift = bn ? bn->ep_shift : nullptr; a.ep_res = bn ? bn->ep_res : nullptr; a.ep_act = bn ? bn->ep_act : 0;
  a.src = src; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = dst; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cs = Cs; a.Cd = Cd;
  const int ncol = wino_use8(mode, Cs, Cd) ? 128 : 64;      // columns per workgroup
  a.Npad = ((Cd + ncol - 1) / ncol) * ncol;
  a.nchunk = (Cs + 31) / 32;
  wino_plan(B, H, W, a.Ph, a.PP, a.tiles_x, a.nx);
  a.ny = a.Npad / ncol;
  a.Hv = H + 1;
  a.magic_hv = div_magic(a.Hv);
  a.magic_pp = div_magic(a.PP);
  a.magic_ny = div_magic1(a.ny);
  a.magic_tx = div_magic1(a.tiles_x);
  const long pack_bytes = HDR + (long)a.nchunk * 24 * (mode == 2 ? 2 : 3) * (((Cd + 127) / 128) * 128) * 16 * 2;
  if (!fs_wino_eligible(mode, B, H, W, Cs, Cd) || pack_bytes >= 2147483647L || (size_t)B * H * W * Cs * 4 >= 4294967000UL ||
      (size_t)B * H * W * Cd * 4 >= 4294967000UL || (long)a.nx * a.ny >= 65536)
    return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)B * H * W * Cs * 4);
  a.dst_bytes = (unsigned)((size_t)B * H * W * Cd * 4);
  a.ws_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  {
    static const int prio = FS_ENV_INT("FS_WINO_PRIO", 0);      // kernel A/B builds only
    a.prio = prio;
  }
  return mode == 2 ? run_wino<PrecF16>(a, w, ws, w_amax, Cin, Cout, transposed, stream)
                   : run_wino<PrecX3>(a, w, ws, w_amax, Cin, Cout, transposed, stream);
}
