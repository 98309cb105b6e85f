// 3x3 / stride-1 / pad-1 convolution (forward and bwd-data) with the minimal-filtering identity F(4,3) applied along the image row
// (round 5): FOUR neighbouring output columns (a "quad") come from 6 transform-domain products per filter row instead of 12 taps,
// i.e. 3 x 6 = 18 MFMA steps per quad = 4.5 per output pixel where the F(2,3) kernel of conv_wino.hip spends 6 and the direct kernel 9.
// The vertical direction stays direct (the 3 filter rows read the same transformed image at shifted rows), so -- unlike F(2x2,3x3),
// which needs 16 accumulator tiles and twice the split work per input element -- the kernel keeps the structure of conv_wino.hip with
// LESS of everything per output pixel: 1.5 transformed values to split (2 there), a 57.6 KB image (76.8 KB there), 3/4 of the MFMAs.
// What it pays: every weight fragment feeds one 32-quad block instead of two 32-pair blocks (B-fragment loads per MFMA double).
//
//   inputs of quad j, row y:  d0..d5 = x[y][4j-1 .. 4j+4]           (interpolation points 0, +-1, +-2, inf)
//     T0 = 4 d0 - 5 d2 + d4              T1 = (d4 - 4 d2) + (d3 - 4 d1)      T2 = (d4 - 4 d2) - (d3 - 4 d1)
//     T3 = (d4 - d2) + 2 (d3 - d1)       T4 = (d4 - d2) - 2 (d3 - d1)        T5 = 4 d1 - 5 d3 + d5
//   filter row (g0 g1 g2):
//     U0 = g0 / 4     U1 = -(g0 + g1 + g2) / 6     U2 = -(g0 - g1 + g2) / 6     U3 = g0 / 24 + g1 / 12 + g2 / 6
//     U4 = g0 / 24 - g1 / 12 + g2 / 6              U5 = g2
//   m_c = sum over the 3 filter rows and the input channels of T_c * U_c        (6 independent GEMMs, fp32 accumulate)
//   out[4j]   = m0 + m1 + m2 + m3 + m4            out[4j+1] = (m1 - m2) + 2 (m3 - m4)
//   out[4j+2] = (m1 + m2) + 4 (m3 + m4)           out[4j+3] = (m1 - m2) + 8 (m3 - m4) + m5
//
// T and U are formed in fp32 and THEN split into the three bf16 planes of the headline mode (conv_split.h), so the products carry
// 24-bit operands.  F(4,3)'s output transform (coefficients 2 / 4 / 8) amplifies the roundings of the accumulators: with one accumulator
// per component the kernel measured 1.3e-6 .. 4.9e-6 of the output maximum against fp64 where F(2,3) gives 3e-7 .. 9e-7; the CPU
// emulation of the pipeline (tools/wino_accuracy.py, profiles/r05/wino_accuracy.txt) reproduces that and shows where it comes from --
// six accumulator roundings per 16-channel group (one per bf16 term): 1.4e-6 .. 3.2e-6, one rounding: 6e-7 .. 9e-7, exact accumulation:
// 2e-7 .. 3e-7.  Hence the two accumulators per component below (leading term / small terms): measured 5e-7 .. 8e-7 forward and
// 1.0e-6 .. 2.2e-6 bwd-data (profiles/r05/wino4_ab.txt), inside the 1e-5 bound every conv test states.  Other interpolation points
// ({0, +-3/4, +-3/2}) would take another third off in the emulation and cost non-power-of-two transform constants; not needed.
//
//   workgroup = 256 threads = 4 waves as 2 (component triples {0,1,2} / {3,4,5}) x 2 (32-column halves); it owns 32 quads
//   (Ph rows x PQ quads = 128 output pixels) x 64 output channels.  A wave holds 3 components x 32 quads x 32 columns (48 accumulator
//   registers).  LDS: 3 planes x 6 components x 40 halo slots x 80 B = 57.6 KB inside a 64 KB region that the epilogue reuses as the
//   exchange buffer; two workgroups per CU.  Batch tiling, persistent tile walk, pack order, epilogue (bias, dropout hash, BatchNorm
//   partial sums, bwd-data extras, inference affine) as in conv_wino.hip: an output pixel is (quad, o = 0..3), the wave of triple t
//   stores outputs 2t and 2t + 1 of its column half.
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;             // 16-bit elements per LDS slot (80 bytes)
constexpr int WNS = 40;             // halo slots per component image: (Ph + 2) * PQ <= WNS
constexpr int NQ = 32;              // quads per workgroup
constexpr int NC = 6;               // transform components
constexpr int CPLANE = WNS * XLD;   // elements per component image
constexpr int PLANE = NC * CPLANE;  // elements per precision plane
constexpr int XCH_BYTES = 65536;    // epilogue exchange: [4 waves][4 outputs][32 rows][32 columns] floats
constexpr int NSTEP = 18;           // MFMA steps of a chunk: 3 filter rows x 3 components x 2 sixteen-channel halves

struct W4Args {
  const float* src; const unsigned char* ws; const float* bias; float* dst; float* stats;
  int B, H, W, Cs, Cd, Npad, nchunk;
  int Ph, PQ, tiles_x, nx, ny, Hv;
  unsigned src_bytes, ws_bytes, dst_bytes;
  unsigned magic_pq, magic_hv, magic_ny, magic_tx;
  const float* bn_y; const unsigned char* bn_mask; const float* bn_mean; const float* bn_invstd;      // bwd-data: BatchNorm-backward sums
  const float* add_src; const unsigned char* add_mask;                                                // bwd-data: residual addend
  const float* ep_scale; const float* ep_shift; const float* ep_res; int ep_act;                      // forward, inference
  float drop_scale; uint32_t drop_thresh, drop_key;
};

// Weight pack: Up[g4 = ((chunk*3 + ky)*6 + c)*2 + s][plane][n][j] = plane-th term of U_c of filter row ky at (k = 32*chunk + 16*s + j, n)
//   forward : row ky = W[ky*3 + 0..2][k][n]                     (K = Cin,  N = Cout)
//   bwd-data: row ky = W[8 - (ky*3 + 0..2)][n][k]               (K = Cout, N = Cin; taps flipped)
template <class P>
__global__ __launch_bounds__(256) void wino4_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, int Cin, int Cout,
                                                         int transposed, int Ks, int Ns, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
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
      g[kx][j] = ok ? v : 0.f;
    }
  }
#pragma unroll
  for (int c = 0; c < NC; ++c) {
    typename P::x8 p[P::NPL];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float g0 = g[0][j], g1 = g[1][j], g2 = g[2][j];
      float u;
      if (c == 0) u = 0.25f * g0;
      else if (c == 1) u = -((g0 + g2) + g1) * (1.f / 6.f);
      else if (c == 2) u = -((g0 + g2) - g1) * (1.f / 6.f);
      else if (c == 3) u = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) + g1 * (1.f / 12.f);
      else if (c == 4) u = (g0 * (1.f / 24.f) + g2 * (1.f / 6.f)) - g1 * (1.f / 12.f);
      else u = g2;
      typename P::T t[P::NPL];
      P::split(u, t);
#pragma unroll
      for (int pl = 0; pl < P::NPL; ++pl) p[pl][j] = t[pl];
    }
    const int g4 = (kyc * NC + c) * 2 + (quarter >> 1);
#pragma unroll
    for (int pl = 0; pl < P::NPL; ++pl)
      *reinterpret_cast<typename P::x8*>(wp + (((long)g4 * P::NPL + pl) * Npad + n) * 16 + 8 * (quarter & 1)) = p[pl];
  }
}

template <class P>
constexpr int wino4_lds_bytes() {
  return XCH_BYTES + 4 * 32 * 2 * 4 /* stats */ + 2 * NQ * 4 /* rowpix x2 */ + 16;
}

template <class P, int RD>
__global__ __launch_bounds__(256, 2) void conv3x3_wino4_kernel(W4Args a) {
  static_assert(!P::SCALED, "bf16x3 only: the fp16 planes of f16x2 have no headroom for the x10 range of the F(4,3) input transform");
  static_assert(NSTEP % RD == 0, "the fragment ring keeps its phase across chunks only if its depth divides the steps of a chunk");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  // (slot, channel quad) items per thread: item 0 = slots 0..31, one per thread; item 1 = slots 32..39, wave 0 only.  (Every wave loading
  // item 1 and splitting a third of its components -- a 1 1/3-pass split phase instead of wave 0's two -- measured +-1 %, 192 -> 192
  // 2 % slower: the re-reads cost what the balance buys; profiles/r05/wino4_ab.txt)
  constexpr int NITEM = 2;
  constexpr int IMG = XCH_BYTES;
  static_assert(NPL * PLANE * 2 <= IMG, "halo image inside the exchange region");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  typename P::T* Ah = reinterpret_cast<typename P::T*>(smem);                  // [NPL][6][WNS][XLD]
  float* red = reinterpret_cast<float*>(smem + IMG);                           // [4 waves][32 columns][2]
  int* rowpix = reinterpret_cast<int*>(smem + IMG + 1024);                     // [2 tile parities][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  // ---- persistent schedule (conv_wino.hip): XCD x owns a contiguous range of tiles, its workgroups take every L-th tile of it ----
  const int ntile = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = ntile >> 3, rm = ntile & 7;
  const int t_first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd);
  const int t_end = t_first + qd + (xcd < rm ? 1 : 0);
  const int L = ((int)gridDim.x + 7 - xcd) >> 3;
  int wg = t_first + loc;
  if (wg >= t_end) return;               // workgroup-uniform

  const int nslots = (a.Ph + 2) * a.PQ, nquads = a.Ph * a.PQ;
  auto image_row = [&](int vy, int& bb, int& yy) {
    const bool in = vy >= 0 && vy < a.B * a.Hv;
    bb = in ? div_small(vy, a.magic_hv) : 0;
    yy = in ? vy - bb * a.Hv : a.H;
  };
  auto decode = [&](int w_, int& mt, int& n0, int& y0, int& x0) {
    mt = div_small1(w_, a.magic_ny);
    n0 = (w_ - mt * a.ny) * 64;
    const int ty = div_small1(mt, a.magic_tx), tx = mt - ty * a.tiles_x;
    y0 = ty * a.Ph; x0 = tx * 4 * a.PQ;
  };
  auto write_rowpix = [&](int par, int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < NQ) {
      const int p = row_perm(t_);
      const int py = div_small(p, a.magic_pq), px = p - py * a.PQ;
      int bb, yy;
      image_row(y0 + py, bb, yy);
      const bool live = p < nquads && yy < a.H && x0 + 4 * px < a.W;
      rowpix[par * NQ + t_] = live ? ((bb * a.H + yy) * a.W + x0 + 4 * px) * a.Cd : -1;      // element offset of the quad's first pixel
    }
  };
  const int q = tid & 7;
  int goff[NITEM], gmask = 0;            // gmask: 6 validity bits per item
  auto setup_loader = [&](int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      const int slot = (t_ >> 3) + 32 * i;
      const int hy = div_small(slot, a.magic_pq), pj = slot - hy * a.PQ;
      const int ix = x0 + 4 * pj - 1;
      int bb, iy;
      image_row(y0 + hy - 1, bb, iy);
      const bool rowok = slot < nslots && iy < a.H;
      int m = 0;
#pragma unroll
      for (int e = 0; e < 6; ++e) m |= (rowok && ix + e >= 0 && ix + e < a.W) ? (1 << e) : 0;
      gmask = i == 0 ? m : (gmask | (m << (6 * i)));
      goff[i] = ((bb * a.H + (rowok ? iy : 0)) * a.W + ix) * a.Cs + 4 * (t_ & 7);
    }
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);

  f32x4 ra[NITEM][6];
  auto load_halo = [&](int chunk) {
    const int c0 = chunk * 32;
    const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      if (i == 1 && wave != 0) continue;                                       // (wave-uniform)
#pragma unroll
      for (int e = 0; e < 6; ++e) {
        const bool ok = cok && ((gmask >> (6 * i + e)) & 1);
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + e * a.Cs + c0) * 4u) : (int)OOB, 0, 0);
        ra[i][e] = __builtin_bit_cast(f32x4, v);
      }
    }
  };
  auto store_halo = [&]() {                // d0..d5 -> T0..T5 -> planes -> image
#pragma unroll
    for (int i = 0; i < NITEM; ++i) {
      if (i == 1 && wave != 0) continue;
      const int slot = (tid >> 3) + 32 * i;
      if (slot >= nslots) continue;
      const f32x4 d0 = ra[i][0], d1 = ra[i][1], d2 = ra[i][2], d3 = ra[i][3], d4 = ra[i][4], d5 = ra[i][5];
      const f32x4 e42 = d4 - 4.f * d2, o31 = d3 - 4.f * d1;
      const f32x4 e2 = d4 - d2, o2 = 2.f * (d3 - d1);
      const f32x4 T[NC] = {(4.f * d0 - 5.f * d2) + d4, e42 + o31, e42 - o31, e2 + o2, e2 - o2, (4.f * d1 - 5.f * d3) + d5};
#pragma unroll
      for (int c = 0; c < NC; ++c) {
        X4 p[NPL];
        P::split4(T[c], p);
        const int o = (c * WNS + slot) * XLD + 4 * q;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ah[pl * PLANE + o]) = p[pl];
      }
    }
  };

  int rowbase;                             // element offset of the wave's first component image, filter row 0 (tile-invariant)
  const int rowstep = a.PQ * XLD;          // elements per halo row (scalar)
  {
    const int p = row_perm(l31);
    const bool live = p < nquads;
    const int py = live ? div_small(p, a.magic_pq) : 0, px = live ? p - py * a.PQ : 0;
    rowbase = 3 * ct * CPLANE + (py * a.PQ + px) * XLD + 8 * lh;
  }
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = a.nchunk * NSTEP;          // B fragments this wave consumes per tile
  auto b_voff = [&](int n0) { return HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2; };

  X8 fa[2][NPL];          // [buffer][plane]
  X8 fb[RD][NPL];         // [ring slot][plane]: fragments run RD - 1 steps ahead of the MFMAs
  auto load_b = [&](int gg, int voff, X8 (&dst)[NPL]) {
    // fragment gg of this wave's stream: (chunk, ky) = gg / 6, (component within the triple, k half) = gg % 6   (wave-uniform)
    const int kyc = (int)(((unsigned)gg * 10923u) >> 16);
    const int g4 = kyc * 12 + 6 * ct + (gg - 6 * kyc);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, g4 * step_bytes + pl * plane_bytes, 0);
      dst[pl] = __builtin_bit_cast(X8, v);
    }
  };
  auto read_a = [&](int step, X8 (&dst)[NPL]) {
    const int ky = step / 6, ci = (step >> 1) % 3, s2 = step & 1;
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      dst[pl] = *reinterpret_cast<const X8*>(&Ah[pl * PLANE + rowbase + ky * rowstep + ci * CPLANE + 16 * s2]);
  };

  // ---- first tile ----
  int mt, n0, y0, x0;
  decode(wg, mt, n0, y0, x0);
  setup_loader(y0, x0);
  int par = 0;
  write_rowpix(par, y0, x0);
  int bvoff = b_voff(n0);
#pragma unroll
  for (int r = 0; r < RD - 1; ++r) load_b(r < G ? r : G - 1, bvoff, fb[r]);
  load_halo(0);

  for (;;) {
    // [component of the triple].  Two accumulators per component: every MFMA rounds its accumulator once, and with six bf16 terms per
    // product it is those roundings -- not the transforms -- that dominate the error (CPU emulation, tools/wino_accuracy.py: six adds per
    // 16-channel group 1.4e-6 .. 3.2e-6 of the output maximum, one add 6e-7 .. 9e-7, exact accumulation 2e-7 .. 3e-7; F(4,3)'s output
    // transform amplifies them by its coefficients 2 / 4 / 8).  So the leading term x1 y1 has `acc` to itself (one rounding per step) and the
    // five small terms (2^-8, 2^-16 of it) meet in `sacc`, whose roundings are 2^-8 smaller; the two are added once per tile.
    f32x16 acc[3], sacc[3];
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[ci][r] = 0.f; sacc[ci][r] = 0.f; }
    int g = 0;
    const int wg_next = wg + L;
    const bool has_next = wg_next < t_end;
    int mt_n = 0, n0_n = 0, y0_n = 0, x0_n = 0, bvoff_n = bvoff;
    if (has_next) {
      decode(wg_next, mt_n, n0_n, y0_n, x0_n);
      bvoff_n = b_voff(n0_n);
    }
    for (int chunk = 0; chunk < a.nchunk; ++chunk) {
      __syncthreads();                      // every wave has finished reading the previous image / exchange buffer
      store_halo();
      __syncthreads();
      read_a(0, fa[0]);
#pragma unroll
      for (int step = 0; step < NSTEP; ++step) {
        if (step + 1 < NSTEP) read_a(step + 1, fa[(step + 1) & 1]);
        if (step == 0 && chunk + 1 < a.nchunk) load_halo(chunk + 1);
        {
          const int gi = g + RD - 1;         // past this tile's last fragment: the next tile's first ones (or a repeat of the last)
          const bool own = gi < G;
          const int gn = gi - G < G ? gi - G : G - 1;
          load_b(own ? gi : (has_next ? gn : G - 1), own ? bvoff : bvoff_n, fb[(step + RD - 1) % RD]);
        }
        __builtin_amdgcn_sched_barrier(0);
        const X8(&A)[NPL] = fa[step & 1];
        const X8(&Bf)[NPL] = fb[step % RD];
        constexpr int dummy = 0; (void)dummy;
        const int ci = (step >> 1) % 3;
#pragma unroll
        for (int t = 0; t + 1 < P::NTERM; ++t) sacc[ci] = P::mfma(A[P::ta(t)], Bf[P::tb(t)], sacc[ci]);   // the small terms, smallest first
        acc[ci] = P::mfma(A[P::ta(P::NTERM - 1)], Bf[P::tb(P::NTERM - 1)], acc[ci]);                      // x1 y1
        __builtin_amdgcn_sched_barrier(0);
        ++g;
      }
    }

    // ---- epilogue: inverse transform across the two component-triple waves through LDS, rows of [pixel][4 channels] out ----
    __syncthreads();                        // the halo image is dead
    {
      // partial sums of the four outputs in accumulator order
      //   triple 0 (m0 m1 m2): m0 + m1 + m2 | m1 - m2 | m1 + m2 | m1 - m2        triple 1 (m3 m4 m5): m3 + m4 | 2 (m3 - m4) | 4 (m3 + m4) | 8 (m3 - m4) + m5
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      float* xw = reinterpret_cast<float*>(smem) + wave * 4096 + (4 * (l_ >> 5)) * 32 + (l_ & 31);      // [output][32 rows][32 columns]
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2);
        const float ma = acc[0][r] + sacc[0][r], mb = acc[1][r] + sacc[1][r], mc = acc[2][r] + sacc[2][r];
        float p0, p1, p2, p3;
        if (ct == 0) {
          const float s = mb + mc, d = mb - mc;
          p0 = ma + s; p1 = d; p2 = s; p3 = d;
        } else {
          const float s = ma + mb, d = ma - mb;
          p0 = s; p1 = 2.f * d; p2 = 4.f * s; p3 = 8.f * d + mc;
        }
        xw[row * 32] = p0;
        xw[1024 + row * 32] = p1;
        xw[2048 + row * 32] = p2;
        xw[3072 + row * 32] = p3;
      }
    }
    // bwd-data extras (BatchNorm-backward sums, residual addend): a four-row window of their operands in flight (conv_wino.hip)
    f32x4 pf_y[4], pf_a[4];
    unsigned pf_ym[4], pf_am[4];
    const bool want_y = a.bn_y != nullptr, want_a = a.add_src != nullptr;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    auto pf_issue = [&](int slot, int i) {   // i = 4 b + k: output 2 ct + b, rows 8 k .. 8 k + 7
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int n_ = n0 + 32 * wn + (l_ & 7) * 4;
      const int pix = rowpix[par * NQ + 8 * (i & 3) + (l_ >> 3)];
      const bool live = pix >= 0 && n_ < a.Cd;
      const unsigned e = (unsigned)(pix + n_ + (2 * ct + (i >> 2)) * a.Cd);
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
    {
      // this wave stores outputs 2 ct and 2 ct + 1 of its 32-column half: lane = (row within a group of 8, channel quad)
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
      const float* mine = reinterpret_cast<const float*>(smem) + wave * 4096 + 2 * ct * 1024;
      const float* theirs = reinterpret_cast<const float*>(smem) + (wave ^ 2) * 4096 + 2 * ct * 1024;
      f32x4 csum = {0.f, 0.f, 0.f, 0.f}, csq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int b = 0; b < 2; ++b)
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const int row = 8 * k + rsub;
          const int pix = rowpix[par * NQ + row];
          const f32x4 m = *reinterpret_cast<const f32x4*>(mine + b * 1024 + row * 32 + c4) + *reinterpret_cast<const f32x4*>(theirs + b * 1024 + row * 32 + c4);
          const bool live = pix >= 0 && nok;
          const unsigned e = (unsigned)(pix + n + (2 * ct + b) * a.Cd);
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
          const int w0 = col >> 5, c31 = col & 31;                          // waves w0 (outputs 0, 1) and w0 + 2 (outputs 2, 3)
          const float v = red[(w0 * 32 + c31) * 2 + which] + red[((w0 + 2) * 32 + c31) * 2 + which];
          if (n0 + col < a.Cd) a.stats[((long)mt * a.Cd + n0 + col) * 2 + which] = v;
        }
      }
    }
    if (!has_next) break;
    wg = wg_next; mt = mt_n; n0 = n0_n; bvoff = bvoff_n; par ^= 1;
  }
}

// ---- eight-wave form (the structure of conv3x3_wino8_kernel, conv_wino.hip): one workgroup of 512 threads per CU owns 32 quads x 128
// columns; waves = 2 component triples x 4 column groups.  Both waves of a SIMD (w and w + 4) stay in the MFMA phase and the split /
// store of the NEXT chunk's halo image is a block of vector code inside the CURRENT chunk's MFMA phase, at step S_EARLY for waves 0-3
// and S_LATE for waves 4-7, into the other of two LDS images (2 x 64 KB: an image is 57.6 KB, an exchange round of the epilogue 64 KB);
// halo loads for chunk j + 2 go into the registers the split of chunk j + 1 just freed.  One barrier per chunk.  The split work per
// MFMA is half the four-wave kernel's (one image feeds 128 columns).  For layers with > 64 destination and >= 256 source channels.
template <class P>
__global__ __launch_bounds__(512, 1) void conv3x3_wino48_kernel(W4Args a) {
  static_assert(!P::SCALED, "bf16x3 only");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  constexpr int IMG = XCH_BYTES;             // bytes reserved per halo image (57.6 KB used) = one exchange round
  constexpr int NCOL = 128;                  // columns per workgroup
  constexpr int S_EARLY = 3, S_LATE = 12;    // steps behind which waves 0-3 / 4-7 split the next chunk
  constexpr int RD = 3;
  static_assert(NPL * PLANE * 2 <= IMG && NSTEP % RD == 0, "image inside its buffer; ring phase kept across chunks");
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  int* rowpix = reinterpret_cast<int*>(smem + 2 * IMG);                        // [2 tile parities][32]

  const int tid = threadIdx.x, lane = tid & 63;
  const int hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ct = hw >> 2;                                                      // component triple; also the stagger group
  const int wn = hw & 3;                                                       // 32-column group
  const int l31 = lane & 31, lh = lane >> 5;

  const int ntile = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = ntile >> 3, rm = ntile & 7;
  const int t_first = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd);
  const int t_end = t_first + qd + (xcd < rm ? 1 : 0);
  const int L = ((int)gridDim.x + 7 - xcd) >> 3;
  if (t_first + loc >= t_end) return;                                          // workgroup-uniform
  const int ntw = (t_end - t_first - loc + L - 1) / L;                         // tiles this workgroup walks: t_first + loc + k L
  const int nch = a.nchunk;
  const int nslots = (a.Ph + 2) * a.PQ, nquads = a.Ph * a.PQ;

  auto image_row = [&](int vy, int& bb, int& yy) {
    const bool in = vy >= 0 && vy < a.B * a.Hv;
    bb = in ? div_small(vy, a.magic_hv) : 0;
    yy = in ? vy - bb * a.Hv : a.H;
  };
  auto decode = [&](int w_, int& mt, int& n0, int& y0, int& x0) {
    mt = div_small1(w_, a.magic_ny);
    n0 = (w_ - mt * a.ny) * NCOL;
    const int ty = div_small1(mt, a.magic_tx), tx = mt - ty * a.tiles_x;
    y0 = ty * a.Ph; x0 = tx * 4 * a.PQ;
  };
  auto write_rowpix = [&](int par, int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    if (t_ < NQ) {
      const int p = row_perm(t_);
      const int py = div_small(p, a.magic_pq), px = p - py * a.PQ;
      int bb, yy;
      image_row(y0 + py, bb, yy);
      const bool live = p < nquads && yy < a.H && x0 + 4 * px < a.W;
      rowpix[par * NQ + t_] = live ? ((bb * a.H + yy) * a.W + x0 + 4 * px) * a.Cd : -1;
    }
  };
  // halo loader: one item per thread = (slot tid >> 3, channel quad tid & 7); slots 0..39 exist: waves 0-4 (wave-uniform)
  const bool has_item = hw < 5;
  int goff = 0, gmask = 0;
  auto setup_loader = [&](int y0, int x0) {
    int t_ = tid;
    asm volatile("" : "+v"(t_));
    const int slot = t_ >> 3;
    const int hy = div_small(slot, a.magic_pq), pj = slot - hy * a.PQ;
    const int ix = x0 + 4 * pj - 1;
    int bb, iy;
    image_row(y0 + hy - 1, bb, iy);
    const bool rowok = slot < nslots && iy < a.H;
    int m = 0;
#pragma unroll
    for (int e = 0; e < 6; ++e) m |= (rowok && ix + e >= 0 && ix + e < a.W) ? (1 << e) : 0;
    gmask = m;
    goff = ((bb * a.H + (rowok ? iy : 0)) * a.W + ix) * a.Cs + 4 * (t_ & 7);
  };
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
  const int q = tid & 7;

  f32x4 ra[6];
  auto load_halo = [&](int chunk) {
    if (!has_item) return;
    const int c0 = chunk * 32;
    const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
    for (int e = 0; e < 6; ++e) {
      const bool ok = cok && ((gmask >> e) & 1);
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff + e * a.Cs + c0) * 4u) : (int)OOB, 0, 0);
      ra[e] = __builtin_bit_cast(f32x4, v);
    }
  };
  auto store_halo = [&](int buf) {         // d0..d5 -> T0..T5 -> planes -> image `buf`
    if (!has_item) return;
    const int slot = tid >> 3;
    if (slot >= nslots) return;
    typename P::T* Aw = reinterpret_cast<typename P::T*>(smem + buf * IMG);
    const f32x4 d0 = ra[0], d1 = ra[1], d2 = ra[2], d3 = ra[3], d4 = ra[4], d5 = ra[5];
    const f32x4 e42 = d4 - 4.f * d2, o31 = d3 - 4.f * d1;
    const f32x4 e2 = d4 - d2, o2 = 2.f * (d3 - d1);
    const f32x4 T[NC] = {(4.f * d0 - 5.f * d2) + d4, e42 + o31, e42 - o31, e2 + o2, e2 - o2, (4.f * d1 - 5.f * d3) + d5};
#pragma unroll
    for (int c = 0; c < NC; ++c) {
      X4 p[NPL];
      P::split4(T[c], p);
      const int o = (c * WNS + slot) * XLD + 4 * q;
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Aw[pl * PLANE + o]) = p[pl];
    }
  };

  int rowbase;
  const int rowstep = a.PQ * XLD;
  {
    const int p = row_perm(l31);
    const bool live = p < nquads;
    const int py = live ? div_small(p, a.magic_pq) : 0, px = live ? p - py * a.PQ : 0;
    rowbase = 3 * ct * CPLANE + (py * a.PQ + px) * XLD + 8 * lh;
  }
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = nch * NSTEP;               // B fragments this wave consumes per tile
  auto b_voff = [&](int n0) { return HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2; };

  X8 fa[2][NPL];
  X8 fb[RD][NPL];
  auto load_b = [&](int gg, int voff, X8 (&dst)[NPL]) {
    const int kyc = (int)(((unsigned)gg * 10923u) >> 16);
    const int g4 = kyc * 12 + 6 * ct + (gg - 6 * kyc);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) {
      const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, voff, g4 * step_bytes + pl * plane_bytes, 0);
      dst[pl] = __builtin_bit_cast(X8, v);
    }
  };
  auto read_a = [&](int buf, int step, X8 (&dst)[NPL]) {
    const int ky = step / 6, ci = (step >> 1) % 3, s2 = step & 1;
    const typename P::T* Ar = reinterpret_cast<const typename P::T*>(smem + buf * IMG);
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl)
      dst[pl] = *reinterpret_cast<const X8*>(&Ar[pl * PLANE + rowbase + ky * rowstep + ci * CPLANE + 16 * s2]);
  };

  // ---- cursors: the halo loader runs up to two chunks ahead of the MFMAs, across tile boundaries ----
  int lk = 0, lc = 0;                      // (tile number, chunk) of the next halo load
  auto loader_advance = [&]() {
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
    f32x16 acc[3], sacc[3];                // leading term / small terms (see the four-wave kernel)
#pragma unroll
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll
      for (int r = 0; r < 16; ++r) { acc[ci][r] = 0.f; sacc[ci][r] = 0.f; }
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
        const X8(&A)[NPL] = fa[st & 1];
        const X8(&Bf)[NPL] = fb[st % RD];
        const int ci = (st >> 1) % 3;
#pragma unroll
        for (int t = 0; t + 1 < P::NTERM; ++t) sacc[ci] = P::mfma(A[P::ta(t)], Bf[P::tb(t)], sacc[ci]);
        acc[ci] = P::mfma(A[P::ta(P::NTERM - 1)], Bf[P::tb(P::NTERM - 1)], acc[ci]);
        __builtin_amdgcn_sched_barrier(0);
        ++g;
        // the NEXT chunk's split / store, at a different step for the two waves of a SIMD (w and w + 4, i.e. ct = 0 / 1)
        if ((st == S_EARLY || st == S_LATE) && split_next && (st == S_EARLY) == (ct == 0)) {
          store_halo(buf ^ 1);
          if (lk < ntw) { load_halo(lc); loader_advance(); }    // chunk J + 2 into the registers just freed
        }
      }
      __syncthreads();
    }

    // ---- epilogue: two exchange rounds through the image buffer the last chunk was read from.  Round r: every wave leaves its partial
    // sums of outputs r and 2 + r; the wave of triple t then stores output 2 t + r of its column group. ----
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
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    f32x4 pf_y[4], pf_a[4];
    unsigned pf_ym[4], pf_am[4];
    auto pf_issue = [&](int slot, int i) {   // i = 4 r + k: output 2 ct + r, rows 8 k .. 8 k + 7
      int l_ = lane;
      asm volatile("" : "+v"(l_));
      const int pix = rowpix[par * NQ + 8 * (i & 3) + (l_ >> 3)];
      const bool live = pix >= 0 && nok;
      const unsigned e = (unsigned)(pix + n + (2 * ct + (i >> 2)) * a.Cd);
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
    for (int r_ = 0; r_ < 2; ++r_) {
      {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        float* xw = reinterpret_cast<float*>(xbuf) + hw * 2048 + (4 * (l_ >> 5)) * 32 + (l_ & 31);      // [wave][slot 0: output r | slot 1: output 2 + r][32 rows][32 columns]
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = (r & 3) + 8 * (r >> 2);
          const float ma = acc[0][r] + sacc[0][r], mb = acc[1][r] + sacc[1][r], mc = acc[2][r] + sacc[2][r];
          float lo, hi;                    // partial sums of outputs r_ and 2 + r_
          if (ct == 0) {
            const float s = mb + mc, d = mb - mc;
            lo = r_ == 0 ? ma + s : d;
            hi = r_ == 0 ? s : d;
          } else {
            const float s = ma + mb, d = ma - mb;
            lo = r_ == 0 ? s : 2.f * d;
            hi = r_ == 0 ? 4.f * s : 8.f * d + mc;
          }
          xw[row * 32] = lo;
          xw[1024 + row * 32] = hi;
        }
      }
      __syncthreads();
      {
        int l_ = lane;
        asm volatile("" : "+v"(l_));
        const int c4 = (l_ & 7) * 4, rsub = l_ >> 3;
        // this wave stores output 2 ct + r_ of its column group: slot ct of itself and of the other triple's wave (hw ^ 4)
        const float* s0 = reinterpret_cast<const float*>(xbuf) + hw * 2048 + ct * 1024;
        const float* s1 = reinterpret_cast<const float*>(xbuf) + (hw ^ 4) * 2048 + ct * 1024;
#pragma unroll
        for (int kk = 0; kk < 4; ++kk) {
          const int row = 8 * kk + rsub;
          const int pix = rowpix[par * NQ + row];
          const f32x4 m = *reinterpret_cast<const f32x4*>(s0 + row * 32 + c4) + *reinterpret_cast<const f32x4*>(s1 + row * 32 + c4);
          const bool live = pix >= 0 && nok;
          const unsigned e = (unsigned)(pix + n + (2 * ct + r_) * a.Cd);
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
          if (want_y) {
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
          if (r_ == 0) pf_issue(kk, 4 + kk);
        }
      }
      __syncthreads();
    }
    if (a.stats != nullptr) {
#pragma unroll
      for (int o = 8; o < 64; o <<= 1)
#pragma unroll
        for (int j = 0; j < 4; ++j) { csum[j] += __shfl_xor(csum[j], o, 64); csq[j] += __shfl_xor(csq[j], o, 64); }
      if (lane < 8 && nok) {               // slab row = (pixel tile, component triple = output pair): [2 nx][Cd][2]
        float* dst = a.stats + ((long)(2 * mt + ct) * a.Cd + n) * 2;
#pragma unroll
        for (int j = 0; j < 4; ++j) { dst[2 * j] = csum[j]; dst[2 * j + 1] = csq[j]; }
      }
    }
    if (!has_next) break;
    mt = mt_n; n0 = n0_n; bvoff = bvoff_n;
    write_rowpix((k + 1) & 1, y0_n, x0_n);
  }
}

// Policy (kernel A/B builds read FS_WINO4; the shipped build carries the constant): 0 off, 1 on where eligible
const int g_wino4 = FS_ENV_INT("FS_WINO4", 1);

// Ph rows x PQ quads <= 32 quads per workgroup, halo (Ph+2)*PQ <= WNS; fewest tiles over the stacked batch, then smallest halo.
void wino4_plan(int B, int H, int W, int& Ph, int& PQ, int& tiles_x, int& nx) {
  const long rows = (long)B * (H + 1);
  const int wq = W / 4;
  long best = -1;
  Ph = 8; PQ = 4;
  for (int pq = 2; pq <= 20 && pq <= (wq < 2 ? 2 : wq); ++pq) {      // div_small needs a divisor >= 2
    int ph = NQ / pq;
    while (ph > 1 && (ph + 2) * pq > WNS) --ph;
    if (ph < 1 || (ph + 2) * pq > WNS) continue;
    const long tiles = (long)cdiv(rows, ph) * cdiv(wq, pq);
    const long cost = tiles * 1000 + (ph + 2) * pq;
    if (best < 0 || cost < best) { best = cost; Ph = ph; PQ = pq; }
  }
  tiles_x = cdiv(wq, PQ);
  nx = cdiv(rows, Ph) * tiles_x;
}

// Which layers take the eight-wave form (profiles/r05/wino48_ab.txt; B = 64, us per call four-wave -> eight-wave): long channel loops whose
// destination channels fill 128-column tiles -- 256 -> 256 @ 20x20 141 -> 130 (fwd) / 142 -> 134 (bwd-data), 960 -> 240 @ 80x80 5 560 -> 5 285,
// its bwd-data (240 -> 960: 960 pads to 1 024) 6 560 -> 5 840; a tie at 128 source channels (128 -> 128 @ 40x40 136 / 136), and a loss
// where 128-column tiles pad the destination by a third (192 -> 192 @ 80x80 980 -> 1 136).
// FS_WINO48 in kernel A/B builds: 0 never, 1 that rule, 2 every layer above 64 destination channels
bool wino4_use8(int Cs, int Cd) {
  static const int pol = FS_ENV_INT("FS_WINO48", 1);
  if (pol == 0 || Cd <= 64) return false;
  const int pad64 = (Cd + 63) / 64 * 64, pad128 = (Cd + 127) / 128 * 128;
  return pol == 2 || (Cs >= 192 && pad128 * 10 <= pad64 * 11);
}

int wino4_grid_slots() {
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

}  // namespace

// bf16x3 only; W a multiple of 4 (>= 8), stacked rows < 65536, channel counts multiples of 4, K >= 32
bool fs_wino4_eligible(int mode, int B, int H, int W, int Cs, int Cd) {
  if (!(g_wino4 != 0 && mode == 1 && W % 4 == 0 && W >= 8 && (long)B * (H + 1) < 65536 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 32)) return false;
  // the tile index is decoded with 16-bit magic divisions: 128-pixel x 64-column tiles must number fewer than 65 536 (configs[3]'s C1 head
  // at 40 images of 160 x 160 x 1024 channels has 129 k of them and stays on the F(2,3) kernels, whose tiles are two to four times larger)
  int Ph, PQ, tx, nx;
  wino4_plan(B, H, W, Ph, PQ, tx, nx);
  const int ncol = wino4_use8(Cs, Cd) ? 128 : 64;
  return (long)nx * ((Cd + ncol - 1) / ncol) < 65536;
}

long fs_wino4_pack_bytes(int mode, int Cs, int Cd) {
  (void)mode;
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 127) / 128) * 128;        // room for either column tiling
  return HDR + nchunk * 36 * 3 * Npad * 16 * 2;
}

int fs_wino4_stats_slabs(int B, int H, int W, int Cs, int Cd) {
  int Ph, PQ, tx, nx;
  wino4_plan(B, H, W, Ph, PQ, tx, nx);
  return wino4_use8(Cs, Cd) ? 2 * nx : nx;        // the eight-wave kernel writes one row per (pixel tile, output pair)
}

int fs_wino4_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, int B, int H, int W, int Cs,
                     int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh, uint32_t drop_key, const FsBnSums* bn,
                     hipStream_t stream) {
  typedef PrecX3 P;
  W4Args a;
  a.bn_y = bn ? bn->y : nullptr; a.bn_mask = bn ? bn->mask : nullptr; a.bn_mean = bn ? bn->mean : nullptr; a.bn_invstd = bn ? bn->invstd : nullptr;
  a.add_src = bn ? bn->add_src : nullptr; a.add_mask = bn ? bn->add_mask : nullptr;
  a.ep_scale = bn ? bn->ep_scale : nullptr; a.ep_shift = bn ? bn->ep_shift : nullptr; a.ep_res = bn ? bn->ep_res : nullptr; a.ep_act = bn ? bn->ep_act : 0;
  a.src = src; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = dst; a.stats = stats;
  a.B = B; a.H = H; a.W = W; a.Cs = Cs; a.Cd = Cd;
  const bool eight = wino4_use8(Cs, Cd);
  const int ncol = eight ? 128 : 64;              // columns per workgroup
  a.Npad = ((Cd + ncol - 1) / ncol) * ncol;
  a.nchunk = (Cs + 31) / 32;
  wino4_plan(B, H, W, a.Ph, a.PQ, a.tiles_x, a.nx);
  a.ny = a.Npad / ncol;
  a.Hv = H + 1;
  a.magic_hv = div_magic(a.Hv);
  a.magic_pq = div_magic(a.PQ);
  a.magic_ny = div_magic1(a.ny);
  a.magic_tx = div_magic1(a.tiles_x);
  const long pack_bytes = fs_wino4_pack_bytes(mode, Cs, Cd);
  if (!fs_wino4_eligible(mode, B, H, W, Cs, Cd) || pack_bytes >= 2147483647L || (size_t)B * H * W * Cs * 4 >= 4294967000UL ||
      (size_t)B * H * W * Cd * 4 >= 4294967000UL || (long)a.nx * a.ny >= 65536 || (long)a.nchunk * NSTEP + 8 >= 16384)
    return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)B * H * W * Cs * 4);
  a.dst_bytes = (unsigned)((size_t)B * H * W * Cd * 4);
  a.ws_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  const long total = (long)a.nchunk * 12 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((wino4_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws),
                       Cin, Cout, transposed, a.Cs, a.Cd, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
  // weight-fragment ring: three slots, fragments two steps ahead (a six-slot ring measured +-1 % on the short-K layers and -11 % on
  // 960 -> 240 with the 60 registers it costs, profiles/r05/wino4_ab.txt)
  constexpr int RD = 3;
  constexpr int lds = wino4_lds_bytes<P>();
  constexpr int lds8 = 2 * XCH_BYTES + 2 * NQ * 4;
  {
    static unsigned long long done = 0ull;            // the dynamic-LDS opt-in is a per-device function attribute
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
    if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
      hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino4_kernel<P, RD>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
      if (attr != hipSuccess) return (int)attr;
      attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_wino48_kernel<P>), hipFuncAttributeMaxDynamicSharedMemorySize, lds8);
      if (attr != hipSuccess) return (int)attr;
      if (dev >= 0 && dev < 64) done |= 1ull << dev;
    }
  }
  const long ntile = (long)a.nx * a.ny;
  const int slots = wino4_grid_slots();
  if (eight) {
    const unsigned grid8 = (unsigned)(ntile < slots / 2 ? ntile : slots / 2);
    hipLaunchKernelGGL((conv3x3_wino48_kernel<P>), dim3(grid8), dim3(512), lds8, stream, a);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  const unsigned grid = (unsigned)(ntile < slots ? ntile : slots);
  hipLaunchKernelGGL((conv3x3_wino4_kernel<P, RD>), dim3(grid), dim3(256), lds, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}
