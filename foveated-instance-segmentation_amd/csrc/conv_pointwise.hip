// 1x1 / stride-1 convolution (forward and bwd-data) = the GEMM  Y[M rows][N] = X[M][K] W[K][N]  in split precision
// (conv_split.h: f16x2 = two scaled fp16 planes / 3 MFMAs per product, bf16x3 = three bf16 planes / 6 MFMAs; fp32 accumulate).
// The Mix-FFN / attention projections of SegFormer, the bottleneck 1x1 convs of HRNet's layer1 and of ResNet-101, HRNet's fuse
// 1x1s.  Same machinery as conv3x3_halo_kernel (conv_halo.hip) without the halo:
//   * a workgroup (4 waves, 2 x 2) owns 128 consecutive rows (pixels) and 64 * NW output channels;
//   * per ROUND it loads, (scales,) splits and writes to LDS TWO 32-channel chunks of its rows (one chunk per round would put
//     three barriers around 12-24 MFMAs per wave: the plain kernel's problem), then runs their 4 k16 steps;
//   * the weights are split ahead of time by a pack kernel into the order the B fragments are consumed,
//     Wp[g = 2 * chunk + s][plane][n][16], so a B fragment is one 16-byte global load per lane straight into the MFMA
//     operand registers, two steps ahead of use;
//   * f16x2 range handling as in the halo kernel: running exponent E over the rounds, accumulators rescaled when it grows,
//     one exponent per weight tensor; result = acc * 2^(E-14) * 2^(Ew-14).
// The plain kernel (conv_igemm_split_kernel) splits both operands inside the kernel for every 32-channel stage and re-reads the
// A tile once per 64 output channels: 60 VALU instructions per MFMA on 64->256 @ 80x80 (profiles/r01/pmc/...shape5.txt).
#include "conv_split.h"
#include "conv_kernels.h"

namespace {

using namespace fs_split;

constexpr int XLD = 40;            // 16-bit elements per LDS row (80 bytes): 32 k + 16 B pad
constexpr int ROWS = 128;
constexpr int RC = 2;              // chunks per round
constexpr int IMG = ROWS * XLD;    // one chunk image of one plane
constexpr int NITEM = 4;           // float4 loads per thread and chunk: rows (tid >> 3) + 32 i, channels 4 (tid & 7)

struct PwArgs {
  const float* src; const unsigned char* ws; const unsigned* ew; const float* bias; float* dst; float* stats;
  long M;
  int Cs, Cd, Npad, nchunk;
  int nx, ny;
  unsigned src_bytes, ws_bytes, dst_bytes;
  float drop_scale; uint32_t drop_thresh, drop_key;
  // gathered rows (forward of a stride >= filter convolution as ONE GEMM with K = taps x Cin): row m = output pixel (b, oy, ox); the
  // K range of tap (r, s) reads input pixel (oy * st + r - pad, ox * st + s - pad), zeros outside the image.  ntap = 0: plain rows.
  int ntap, S, Hx, Wx, Ho, Wo, st, pad, Cin, rounds_per_tap;
  // scattered rows (bwd-data of the same layers, one launch per tap): row m = (b, oy, ox) of the dY grid extended to sc_Ho' x sc_Wo' =
  // (Ho, Wo above; rows past the real grid read zeros) is written to dX pixel (oy * sc_st + sc_dy, ox * sc_st + sc_dx) when that is inside
  // sc_H x sc_W: every pixel of the tap's residue class is written exactly once, the classes no tap reaches are zero-filled by the caller.
  int sc_on, sc_st, sc_dy, sc_dx, sc_H, sc_W;
  // bwd-data extras (round 5; plain rows only): BatchNorm-backward column sums of the layer whose output gradient dst is -- stats then holds
  // (sum dz, sum dz * zhat) with dz = dst masked by that layer's activation bits -- and a second gradient added to dst in the same epilogue
  const float* bn_y; const unsigned char* bn_mask; const float* bn_mean; const float* bn_invstd;
  const float* add_src; const unsigned char* add_mask;
  // forward with a residual (FsBnSums, conv_kernels.h): dst = add_src + DropPath(Dropout(acc + bias)); dp_rows rows per sample (>= ROWS)
  float dp_scale; uint32_t dp_thresh, dp_key; int dp_rows;
};

// Wp[g = 2 * chunk + s][plane][n][j] = plane-th term of Wt[k = 32 * chunk + 16 * s + j][n] (scaled by 2^(14-Ew) in f16x2), behind a
// HDR-byte header.  forward: Wt[k][n] = W[k][n] (K = Cin, N = Cout); bwd-data: Wt[k][n] = W[n][k] (K = Cout, N = Cin).
template <class P>
__global__ __launch_bounds__(256) void conv1x1_pack_kernel(const float* __restrict__ w, unsigned char* __restrict__ ws, const unsigned* __restrict__ ew,
                                                           int Cin, int Cout, int transposed, int Ks, int Ns, int Npad, long total) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  (void)Cin;
  const float sc = P::SCALED ? pow2f(14 - exponent_of_bits(*ew)) : 1.f;
  typename P::T* wp = reinterpret_cast<typename P::T*>(ws + HDR);
  const int n = (int)(idx % Npad);
  const int g = (int)(idx / Npad);
  const int k0 = g * 16;
  typename P::x8 p[P::NPL][2];
#pragma unroll
  for (int j = 0; j < 16; ++j) {
    const int k = k0 + j;
    float v = 0.f;
    if (n < Ns && k < Ks) v = transposed ? w[(long)n * Cout + k] : w[(long)k * Cout + n];
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

// DB (bf16x3, kernel A/B builds only -- measured slower, see run_pointwise): the two LDS chunk images are a double buffer -- while the
// MFMAs of 32-channel chunk c run on one image, chunk c + 1 (loaded one round earlier) is split and stored into the other, half of it
// behind each k16 step; ONE barrier per chunk instead of load -> barrier -> split + store -> barrier -> MFMAs per pair of chunks.
template <class P, int NW, bool DB, bool BN = false>
__global__ __launch_bounds__(256, 2) void conv1x1_gemm_kernel(PwArgs a) {
  static_assert(!DB || !P::SCALED, "the double-buffered loop has no running-exponent step");
  static_assert(!(DB && BN), "the bwd-data extras live in the shipped loop only");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  __shared__ __attribute__((aligned(16))) typename P::T Ah[NPL * RC * IMG];
  __shared__ unsigned amax_cell[2];

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int l31 = lane & 31, lh = lane >> 5;
  const int nwg = a.nx * a.ny;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rm = nwg & 7;
  const int wg = (xcd < rm ? xcd * (qd + 1) : rm * (qd + 1) + (xcd - rm) * qd) + loc;      // XCD-aware: the ny channel tiles of a row tile share an L2
  const int mt = wg / a.ny;
  const int n0 = (wg - mt * a.ny) * 64 * NW;
  const long m0 = (long)mt * ROWS;

  if (tid < 2) amax_cell[tid] = 0u;
  const int q = tid & 7;
  long goff[NITEM];          // element offset of (row, channel quad) in the source, -1 past the last row
  int giy[NITEM], gix[NITEM], gb[NITEM];      // gathered rows: input pixel of tap (0, 0) and the image's first input row
#pragma unroll
  for (int i = 0; i < NITEM; ++i) {
    const long row = m0 + (tid >> 3) + 32 * i;
    goff[i] = row < a.M ? row * a.Cs + 4 * q : -1;
    giy[i] = gix[i] = gb[i] = 0;
    if (a.ntap > 0 && row < a.M) {
      const int ox = (int)(row % a.Wo), t = (int)(row / a.Wo), oy = t % a.Ho, b = t / a.Ho;
      giy[i] = oy * a.st - a.pad; gix[i] = ox * a.st - a.pad; gb[i] = b * a.Hx;
    }
  }
  const __amdgpu_buffer_rsrc_t rsrc_a = make_rsrc(a.src, a.src_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_w = make_rsrc(a.ws, a.ws_bytes);

  f32x4 ra[RC][NITEM];
  auto load_round = [&](int round) {
    if (a.ntap > 0) {          // a round (64 channels) lies inside one tap: Cin is a multiple of 64 on this route
      const int tap = round / a.rounds_per_tap, cbase = (round - tap * a.rounds_per_tap) * (32 * RC);
      const int r = tap / a.S, s_ = tap - r * a.S;
#pragma unroll
      for (int i = 0; i < NITEM; ++i) {
        const int iy = giy[i] + r, ix = gix[i] + s_;
        const bool ok = goff[i] >= 0 && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
        const unsigned base = ((unsigned)((gb[i] + iy) * a.Wx + ix) * (unsigned)a.Cin + (unsigned)(cbase + 4 * q)) * 4u;
#pragma unroll
        for (int cc = 0; cc < RC; ++cc) {
          const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)(base + (unsigned)cc * 128u) : (int)OOB, 0, 0);
          ra[cc][i] = __builtin_bit_cast(f32x4, v);
        }
      }
      return;
    }
#pragma unroll
    for (int cc = 0; cc < RC; ++cc) {
      const int c0 = (round * RC + cc) * 32;
      const bool cok = c0 + 4 * q < a.Cs;
#pragma unroll
      for (int i = 0; i < NITEM; ++i) {
        const bool ok = cok && goff[i] >= 0;
        const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + c0) * 4u) : (int)OOB, 0, 0);
        ra[cc][i] = __builtin_bit_cast(f32x4, v);
      }
    }
  };
  auto tile_amax = [&](int cell) {
    float m = 0.f;
#pragma unroll
    for (int cc = 0; cc < RC; ++cc)
#pragma unroll
      for (int i = 0; i < NITEM; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) m = fmaxf(m, fabsf(ra[cc][i][e]));
    m = wave_max(m);
    if (lane == 0) atomicMax(&amax_cell[cell], __builtin_bit_cast(unsigned, m));
  };
  auto store_round = [&](float sc) {
#pragma unroll
    for (int cc = 0; cc < RC; ++cc)
#pragma unroll
      for (int i = 0; i < NITEM; ++i) {
        const int row = (tid >> 3) + 32 * i;
        X4 p[NPL];
        P::split4(P::SCALED ? ra[cc][i] * sc : ra[cc][i], p);
        const int o = cc * IMG + row * XLD + 4 * q;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ah[pl * RC * IMG + o]) = p[pl];
      }
  };

  int rowbase[2];          // LDS element offset of this lane's A rows (two 32-row tiles of the wave's 64 rows)
#pragma unroll
  for (int mi = 0; mi < 2; ++mi) rowbase[mi] = (64 * wm + 32 * mi + row_perm(l31)) * XLD + 8 * lh;
  const int bvoff = HDR + ((n0 + 32 * wn + l31) * 16 + 8 * lh) * 2;      // sub-tile j: + j * 64 columns = j * 2048 bytes
  const int plane_bytes = a.Npad * 32;
  const int step_bytes = NPL * plane_bytes;
  const int G = a.nchunk * 2;                                             // k16 steps that carry data
  const int nround = (a.nchunk + RC - 1) / RC;

  X8 fa[2][2][NPL];       // [buffer][mi][plane]
  X8 fb[3][NW][NPL];      // ring: fragments run 2 steps ahead of the MFMAs (4 steps per round: 4 % 3 != 0, so the slot is g % 3)
  auto load_b = [&](int g, X8 (&dst)[NW][NPL]) {
    const int gg = g < G ? g : G - 1;            // steps past the last chunk multiply zeros of A: any finite B will do
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
  if constexpr (DB) {
    f32x4 rq[2][NITEM];          // chunk c + 1 (being split) and chunk c + 2 (in flight)
    auto load_chunk = [&](int c, f32x4 (&dst)[NITEM]) {
      const bool in_k = c < a.nchunk;
      if (a.ntap > 0) {
        const int cpt = RC * a.rounds_per_tap, tap = c / cpt, cbase = (c - tap * cpt) * 32;
        const int r = tap / a.S, s_ = tap - r * a.S;
#pragma unroll
        for (int i = 0; i < NITEM; ++i) {
          const int iy = giy[i] + r, ix = gix[i] + s_;
          const bool ok = in_k && goff[i] >= 0 && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
          const unsigned base = ((unsigned)((gb[i] + iy) * a.Wx + ix) * (unsigned)a.Cin + (unsigned)(cbase + 4 * q)) * 4u;
          dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)base : (int)OOB, 0, 0));
        }
        return;
      }
      const int c0 = c * 32;
      const bool cok = in_k && c0 + 4 * q < a.Cs;
#pragma unroll
      for (int i = 0; i < NITEM; ++i) {
        const bool ok = cok && goff[i] >= 0;
        dst[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_a, ok ? (int)((unsigned)(goff[i] + c0) * 4u) : (int)OOB, 0, 0));
      }
    };
    auto store_items = [&](const f32x4 (&src)[NITEM], int buf, int i0, int i1) {
#pragma unroll
      for (int i = i0; i < i1; ++i) {
        X4 p[NPL];
        P::split4(src[i], p);
        const int o = buf * NPL * IMG + ((tid >> 3) + 32 * i) * XLD + 4 * q;
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Ah[o + pl * IMG]) = p[pl];
      }
    };
    load_chunk(0, rq[0]);
    load_chunk(1, rq[1]);
    store_items(rq[0], 0, 0, NITEM);
    load_chunk(2, rq[0]);
    load_b(0, fb[0]); load_b(1, fb[1]);
    __syncthreads();
    // six chunks per trip: buffer, register set and B ring slot are compile-time constants (6 even, 12 steps % 3 == 0)
    for (int c6 = 0; c6 < a.nchunk; c6 += 6) {
#pragma unroll
      for (int cc = 0; cc < 6; ++cc) {
        const int c = c6 + cc;
        if (c < a.nchunk) {
          const int bo = (cc & 1) * NPL * IMG;
#pragma unroll
          for (int mi = 0; mi < 2; ++mi)
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[0][mi][pl] = *reinterpret_cast<const X8*>(&Ah[bo + pl * IMG + rowbase[mi]]);
#pragma unroll
          for (int step = 0; step < 2; ++step) {
            const int gl = cc * 2 + step;
            if (step == 0) {
#pragma unroll
              for (int mi = 0; mi < 2; ++mi)
#pragma unroll
                for (int pl = 0; pl < NPL; ++pl) fa[1][mi][pl] = *reinterpret_cast<const X8*>(&Ah[bo + pl * IMG + rowbase[mi] + 16]);
            }
            load_b(c * 2 + step + 2, fb[(gl + 2) % 3]);
            __builtin_amdgcn_sched_barrier(0);
            const X8(&A)[2][NPL] = fa[step];
            const X8(&Bf)[NW][NPL] = fb[gl % 3];
#pragma unroll
            for (int j = 0; j < NW; ++j)
#pragma unroll
              for (int t = 0; t < P::NTERM; ++t) {
                acc[0][j] = P::mfma(A[0][P::ta(t)], Bf[j][P::tb(t)], acc[0][j]);
                acc[1][j] = P::mfma(A[1][P::ta(t)], Bf[j][P::tb(t)], acc[1][j]);
              }
            store_items(rq[(cc + 1) & 1], (cc + 1) & 1, 2 * step, 2 * step + 2);      // (past the last chunk: zeros into a dead image)
            // ask for the split between the MFMAs, not behind them: 2 MFMAs, 6 vector instructions, and an LDS store every other pair
#pragma unroll
            for (int i = 0; i < 6 * NW; ++i) {
              __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
              __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
              if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
          }
          __syncthreads();
          load_chunk(c + 3, rq[(cc + 1) & 1]);
        }
      }
    }
  } else {
  load_round(0);
  __syncthreads();                        // amax cells zeroed before the first atomic
  // rounds are processed three at a time so that the B ring slot (g % 3) is a compile-time constant: 3 rounds = 12 steps
  for (int r3 = 0; r3 < nround; r3 += 3) {
    if (r3 == 0) { load_b(0, fb[0]); load_b(1, fb[1]); }
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
      const int round = r3 + rr;
      if (round < nround) {
        if (P::SCALED) tile_amax(round & 1);
        __syncthreads();                      // amax complete; every wave has finished reading the previous round's images
        if (P::SCALED) {
          const int ec = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[round & 1]));
          if (ec > E) {
            if (round > 0) {
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
          if (tid == 0) amax_cell[(round + 1) & 1] = 0u;
        }
        store_round(pow2f(14 - E));
        __syncthreads();
        if (round + 1 < nround) load_round(round + 1);
#pragma unroll
        for (int mi = 0; mi < 2; ++mi)
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) fa[0][mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * RC * IMG + rowbase[mi]]);
#pragma unroll
        for (int step = 0; step < 2 * RC; ++step) {
          const int gl = rr * 2 * RC + step;             // step index within the 3-round group: compile-time
          if (step + 1 < 2 * RC) {
            const int cc = (step + 1) >> 1, s2 = (step + 1) & 1;
#pragma unroll
            for (int mi = 0; mi < 2; ++mi)
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl)
                fa[(step + 1) & 1][mi][pl] = *reinterpret_cast<const X8*>(&Ah[pl * RC * IMG + cc * IMG + rowbase[mi] + 16 * s2]);
          }
          load_b(round * 2 * RC + step + 2, fb[(gl + 2) % 3]);
          __builtin_amdgcn_sched_barrier(0);
          const X8(&A)[2][NPL] = fa[step & 1];
          const X8(&Bf)[NW][NPL] = fb[gl % 3];
#pragma unroll
          for (int j = 0; j < NW; ++j)
#pragma unroll
            for (int t = 0; t < P::NTERM; ++t) {
              acc[0][j] = P::mfma(A[0][P::ta(t)], Bf[j][P::tb(t)], acc[0][j]);
              acc[1][j] = P::mfma(A[1][P::ta(t)], Bf[j][P::tb(t)], acc[1][j]);
            }
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }
  }

  // ---- epilogue ----
  __syncthreads();
  float f1 = 1.f, f2 = 1.f;
  if (P::SCALED) {
    const int Ew = exponent_of_bits(*a.ew);
    const int es = E + Ew - 28;
    const bool one = es >= -126 && es <= 127;
    f1 = one ? pow2f(es) : pow2f(E - 14);
    f2 = one ? 1.f : pow2f(Ew - 14);
  }
  const __amdgpu_buffer_rsrc_t rsrc_d = make_rsrc(a.dst, a.dst_bytes);
  int* const rowtab = reinterpret_cast<int*>(&Ah[0]);      // scattered rows: element offset of each of the 128 rows' pixel, -1 = no pixel
  if (a.sc_on) {
    if (tid < ROWS) {
      const long row = m0 + tid;
      int off = -1;
      if (row < a.M) {
        const int ox = (int)(row % a.Wo), t = (int)(row / a.Wo), oy = t % a.Ho, b = t / a.Ho;
        const int iy = oy * a.sc_st + a.sc_dy, ix = ox * a.sc_st + a.sc_dx;
        if (iy >= 0 && iy < a.sc_H && ix >= 0 && ix < a.sc_W) off = ((b * a.sc_H + iy) * a.sc_W + ix) * a.Cd;
      }
      rowtab[tid] = off;
    }
    __syncthreads();
  }
  float csum[NW], csq[NW];
  if constexpr (BN) {
    // The extras' operands sit at the addresses of the stores (lane = column, register = row): one group of 8 rows (half of tile mi of sub-tile
    // j) is requested while the group before it is stored, so a round trip is covered by 8 stores instead of standing in front of each.
    // An absent operand gets a zero-size descriptor: its loads return 0 without touching memory (the window stays branch-free).
    const bool want_y = a.bn_y != nullptr, want_a = a.add_src != nullptr;
    const __amdgpu_buffer_rsrc_t rs_y = make_rsrc(a.bn_y, want_y ? a.dst_bytes : 0u), rs_a = make_rsrc(a.add_src, want_a ? a.dst_bytes : 0u);
    const __amdgpu_buffer_rsrc_t rs_ym = make_rsrc(a.bn_mask, a.bn_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const __amdgpu_buffer_rsrc_t rs_am = make_rsrc(a.add_mask, a.add_mask != nullptr ? a.dst_bytes >> 4 : 0u);
    const unsigned ym_all = a.bn_mask != nullptr ? 0u : 0xFu, am_all = a.add_mask != nullptr ? 0u : 0xFu;
    // DropPath of the forward-with-residual form: a 128-row tile spans at most two samples (dp_rows >= ROWS)
    float dpk0 = 1.f, dpk1 = 1.f;
    long dp_edge = a.M;
    if (a.dp_thresh != 0u) {
      const uint32_t b0 = (uint32_t)(m0 / a.dp_rows);
      dp_edge = (long)(b0 + 1u) * a.dp_rows;
      dpk0 = fs_dropout_keep(b0, a.dp_key, a.dp_thresh) ? a.dp_scale : 0.f;
      dpk1 = fs_dropout_keep(b0 + 1u, a.dp_key, a.dp_thresh) ? a.dp_scale : 0.f;
    }
    float yv[2][8], av[2][8];
    unsigned ymv[2][8], amv[2][8];
    auto element = [&](int g, int k, unsigned& e) -> bool {            // group g = (sub-tile j, tile mi, row half): 8 accumulator rows
      const int j = g >> 2, mi = (g >> 1) & 1, r = 8 * (g & 1) + k;
      const int n = n0 + 64 * j + 32 * wn + l31;
      const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
      const long row = m0 + 64 * wm + 32 * mi + row_perm(trow);
      e = (unsigned)(row * a.Cd + n);
      return row < a.M && n < a.Cd;
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
    issue(0, 0);
#pragma unroll
    for (int g = 0; g < 4 * NW; ++g) {
      const int j = g >> 2, mi = (g >> 1) & 1, slot = g & 1;
      if (g + 1 < 4 * NW) issue(g + 1, slot ^ 1);
      const int n = n0 + 64 * j + 32 * wn + l31;
      const bool nok = n < a.Cd;
      if ((g & 3) == 0) { csum[j] = 0.f; csq[j] = 0.f; }
      const float bv = (a.bias != nullptr && nok) ? a.bias[n] : 0.f;
      const float mu = (want_y && nok) ? a.bn_mean[n] : 0.f, is = (want_y && nok) ? a.bn_invstd[n] : 0.f;
      const unsigned bit = (unsigned)(n & 3);
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        const int r = 8 * (g & 1) + k;
        unsigned e;
        const bool live = element(g, k, e);
        float v = P::SCALED ? fmaf(acc[mi][j][r] * f2, f1, bv) : acc[mi][j][r] + bv;
        if (a.drop_thresh != 0u) v = fs_dropout_keep((uint32_t)e, a.drop_key, a.drop_thresh) ? v * a.drop_scale : 0.f;
        if (a.dp_thresh != 0u) {
          const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
          v *= (m0 + 64 * wm + 32 * mi + row_perm(trow)) < dp_edge ? dpk0 : dpk1;
        }
        if (want_a) v += ((amv[slot][k] >> bit) & 1u) ? av[slot][k] : 0.f;
        v = live ? v : 0.f;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rsrc_d, live ? (int)(e * 4u) : (int)OOB, 0, 0);
        if (want_y) {
          const float dz = ((ymv[slot][k] >> bit) & 1u) ? v : 0.f;
          csum[j] += dz; csq[j] += dz * ((yv[slot][k] - mu) * is);
        } else {
          csum[j] += v; csq[j] += v * v;
        }
      }
    }
  } else {
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    csum[j] = 0.f; csq[j] = 0.f;
    const int n = n0 + 64 * j + 32 * wn + l31;
    if (n >= a.Cd) continue;
    const float bv = (a.bias != nullptr) ? a.bias[n] : 0.f;
#pragma unroll
    for (int mi = 0; mi < 2; ++mi) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        // accumulator row r of tile mi = tile row (r&3) + 8 (r>>2) + 4 lh; LDS row permutation: tile row t holds pixel row_perm(t)
        const int trow = (r & 3) + 8 * (r >> 2) + 4 * lh;
        const long row = m0 + 64 * wm + 32 * mi + row_perm(trow);
        const int ro = a.sc_on ? rowtab[64 * wm + 32 * mi + row_perm(trow)] : 0;
        const bool live = a.sc_on ? ro >= 0 : row < a.M;
        const unsigned e = a.sc_on ? (unsigned)(ro + n) : (unsigned)(row * a.Cd + n);      // element index (< 2^30: dst_bytes < 4 GB)
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
    float* red = reinterpret_cast<float*>(&Ah[0]);     // [wm][64*NW cols][2]; the operand images are dead since the barrier above
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

static inline int pw_nw(int Cd) { return Cd >= 128 ? 2 : 1; }

template <class P>
static int run_pointwise(PwArgs& a, const float* w, void* ws, const unsigned* w_amax, int Cin, int Cout, int transposed, int nw, hipStream_t stream) {
  int e = FS_OK;
  a.ew = P::SCALED ? fs_f16_weight_amax(w, (long)Cin * Cout, ws, w_amax, stream, &e) : nullptr;
  if (e != FS_OK) return e;
  const long total = (long)a.nchunk * 2 * a.Npad;
  if (fs_ws_mode_tls != FS_WS_RUN_ONLY) {
    hipLaunchKernelGGL((conv1x1_pack_kernel<P>), dim3((unsigned)((total + 255) / 256)), dim3(256), 0, stream, w, reinterpret_cast<unsigned char*>(ws), a.ew,
                       Cin, Cout, transposed, a.Cs, a.Cd, a.Npad, total);
    FS_LAUNCH_CHECK();
  }
  if (fs_ws_mode_tls == FS_WS_PACK_ONLY) return FS_OK;
#ifdef FS_EXPERIMENTS
  // the double-buffered chunk loop (bf16x3), measured and rejected (profiles/r04/pw_db_ab.txt: 960 -> 512 3x3 s4 forward 1 273 -> 1 360 us,
  // configs[3] -1.2 %, configs[4] -4 % with the split + store behind each step's 24 MFMAs; 1 225 -> 1 284 us, -2 %, -3 % with it interleaved
  // between the MFMAs by sched_group_barrier, the form kept here): two workgroups per CU alternating whole phases beat one wave
  // interleaving its own.  Only the A/B build carries it (FS_PW_DB=1 from three chunks up, 2 always)
  static const int db_pol = FS_ENV_INT("FS_PW_DB", 0);
  const bool db = !P::SCALED && (db_pol == 2 || (db_pol == 1 && a.nchunk >= 3));
  if constexpr (!P::SCALED) {
    if (db) {
      if (nw == 2) hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 2, true>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
      else hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 1, true>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
      FS_LAUNCH_CHECK();
      return FS_OK;
    }
  }
#endif
  if (a.bn_y != nullptr || a.add_src != nullptr) {
    if (nw == 2) hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 2, false, true>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 1, false, true>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
  if (nw == 2) hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 2, false>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  else hipLaunchKernelGGL((conv1x1_gemm_kernel<P, 1, false>), dim3((unsigned)(a.nx * a.ny)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

}  // namespace

bool fs_pointwise_eligible(int Cs, int Cd, int R, int S, int stride, int pad, int dil) {
  return R == 1 && S == 1 && stride == 1 && pad == 0 && dil == 1 && Cs % 4 == 0 && Cd % 4 == 0 && Cs >= 32;
}

long fs_pointwise_pack_bytes(int mode, int Cs, int Cd) {
  const int nw = pw_nw(Cd), npl = mode == 2 ? 2 : 3;
  const long nchunk = (Cs + 31) / 32, Npad = ((Cd + 64 * nw - 1) / (64 * nw)) * 64 * nw;
  return HDR + nchunk * 2 * npl * Npad * 16 * 2;
}

// rows of the [rows][Cd][2] slab the epilogue's column sums go to: one per 128-row tile
int fs_pointwise_stats_slabs(long M) { return (int)cdiv(M, ROWS); }

int fs_pointwise_conv(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                      long M, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh, uint32_t drop_key,
                      const FsBnSums* bn, hipStream_t stream) {
  PwArgs a;
  a.src = src; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = dst; a.stats = stats;
  if (bn != nullptr && bn->y != nullptr && (!transposed || stats == nullptr)) return FS_ERR_ARG;      // the sums are a bwd-data extra
  if (bn != nullptr && bn->dp_thresh != 0u && (transposed || bn->add_src == nullptr || bn->dp_rows < ROWS)) return FS_ERR_ARG;
  a.bn_y = bn ? bn->y : nullptr; a.bn_mask = bn ? bn->mask : nullptr; a.bn_mean = bn ? bn->mean : nullptr; a.bn_invstd = bn ? bn->invstd : nullptr;
  a.add_src = bn ? bn->add_src : nullptr; a.add_mask = bn ? bn->add_mask : nullptr;
  a.dp_scale = bn ? bn->dp_scale : 1.f; a.dp_thresh = bn ? bn->dp_thresh : 0u; a.dp_key = bn ? bn->dp_key : 0u; a.dp_rows = bn ? bn->dp_rows : 0;
  a.M = M; a.Cs = Cs; a.Cd = Cd;
  const int nwp = pw_nw(Cd);
  a.Npad = ((Cd + 64 * nwp - 1) / (64 * nwp)) * 64 * nwp;
  a.nchunk = (Cs + 31) / 32;
  a.nx = cdiv(M, ROWS);
  // 128-column workgroups (two sub-tiles per wave, the A tile split once for twice the MFMAs) or 64-column ones: whichever needs less
  // time by rounds of 512 resident workgroups x cost of a workgroup per chunk (a refill ~ 1 000 cycles + 768 cycles of MFMAs per 64 columns).
  // N = 320 (the Mix-Transformer's stage-3 width) is the case that matters: 128-column tiles pad it to 384 and leave 600 workgroups =
  // 1.17 rounds, 64-column tiles give 1 000 workgroups of 0.7x the work each (1280 -> 320 forward: 129 TF against 163 for 320 -> 1280).
  int nw = 1;
  if (nwp == 2) {
    const long wg2 = (long)a.nx * (a.Npad / 128), wg1 = (long)a.nx * ((Cd + 63) / 64);
    const long cost2 = ((wg2 + 511) / 512) * (1000 + 2 * 768), cost1 = ((wg1 + 511) / 512) * (1000 + 768);
    static const int force = FS_ENV_INT("FS_PW_NW", 0);      // kernel A/B builds only: 1 / 2 force the tiling
    nw = force == 1 ? 1 : (force == 2 ? 2 : ((wg2 >= 440 && cost2 <= cost1) ? 2 : 1));
  }
  a.ny = nw == 2 ? a.Npad / 128 : (Cd + 63) / 64;
  const long pack_bytes = fs_pointwise_pack_bytes(mode, Cs, Cd);
  if (pack_bytes >= 2147483647L || (size_t)M * Cs * 4 >= 4294967000UL || (size_t)M * Cd * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)M * Cs * 4);
  a.dst_bytes = (unsigned)((size_t)M * Cd * 4);
  a.ws_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  a.ntap = 0; a.S = 1; a.Hx = a.Wx = a.Ho = a.Wo = 1; a.st = 1; a.pad = 0; a.Cin = Cs; a.rounds_per_tap = 1;
  a.sc_on = 0; a.sc_st = 1; a.sc_dy = a.sc_dx = 0; a.sc_H = a.sc_W = 1;
  return mode == 2 ? run_pointwise<PrecF16>(a, w, ws, w_amax, Cin, Cout, transposed, nw, stream)
                   : run_pointwise<PrecX3>(a, w, ws, w_amax, Cin, Cout, transposed, nw, stream);
}

// Forward of a convolution whose stride is at least its filter size (HRNet's C1 classification branch: 960 -> 512, 3x3, stride 4; the
// 1x1 stride-4 shortcut): no input pixel is shared between outputs, so the layer IS the GEMM  Y[B*Ho*Wo][Cout] = A[.][R*S*Cin] W[R*S*Cin][Cout]
// with the A rows gathered per tap -- the weights in RSCK order are already that W.  The plain kernel this replaces split the A tile once
// per 64-column tile (eight times for 512 output channels: 1.96 ms, 115 TF on 960 -> 512 @ 80x80, B = 64).
bool fs_pointwise_gather_eligible(int Cin, int Cout, int R, int S, int stride, int dil) {
  return stride >= R && stride >= S && stride > 1 && dil == 1 && Cin % 64 == 0 && Cout % 4 == 0 && R * S <= 9;
}

int fs_pointwise_gather_conv(int mode, const float* x, const float* w, const float* bias, float* y, float* stats, void* ws, const unsigned* w_amax,
                             int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, float drop_scale,
                             uint32_t drop_thresh, uint32_t drop_key, hipStream_t stream) {
  if (!fs_pointwise_gather_eligible(Cin, Cout, R, S, stride, 1)) return FS_ERR_ARG;
  PwArgs a;
  const int K = R * S * Cin;
  a.src = x; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = bias; a.dst = y; a.stats = stats;
  a.bn_y = nullptr; a.bn_mask = nullptr; a.bn_mean = nullptr; a.bn_invstd = nullptr; a.add_src = nullptr; a.add_mask = nullptr;
  a.dp_scale = 1.f; a.dp_thresh = 0u; a.dp_key = 0u; a.dp_rows = 0;
  a.M = (long)B * Ho * Wo; a.Cs = K; a.Cd = Cout;
  const int nwp = pw_nw(Cout);
  a.Npad = ((Cout + 64 * nwp - 1) / (64 * nwp)) * 64 * nwp;
  a.nchunk = K / 32;
  a.nx = cdiv(a.M, ROWS);
  const int nw = nwp;
  a.ny = nw == 2 ? a.Npad / 128 : (Cout + 63) / 64;
  const long pack_bytes = fs_pointwise_pack_bytes(mode, K, Cout);
  if (pack_bytes >= 2147483647L || (size_t)B * H * W * Cin * 4 >= 4294967000UL || (size_t)a.M * Cout * 4 >= 4294967000UL) return FS_ERR_ARG;
  a.src_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
  a.dst_bytes = (unsigned)((size_t)a.M * Cout * 4);
  a.ws_bytes = (unsigned)pack_bytes;
  a.drop_scale = drop_scale; a.drop_thresh = drop_thresh; a.drop_key = drop_key;
  a.ntap = R * S; a.S = S; a.Hx = H; a.Wx = W; a.Ho = Ho; a.Wo = Wo; a.st = stride; a.pad = pad; a.Cin = Cin; a.rounds_per_tap = Cin / (32 * RC);
  a.sc_on = 0; a.sc_st = 1; a.sc_dy = a.sc_dx = 0; a.sc_H = a.sc_W = 1;
  return mode == 2 ? run_pointwise<PrecF16>(a, w, ws, w_amax, K, Cout, 0, nw, stream)
                   : run_pointwise<PrecX3>(a, w, ws, w_amax, K, Cout, 0, nw, stream);
}

// bwd-data of the same layers: an input pixel receives at most ONE tap, so dX = nine (R*S) GEMMs  dY[rows][Cout] W_tap^T[Cout][Cin]  whose
// result rows land on the tap's residue class of dX.  One launch per tap (each packs its W_tap^T into ws; the launches are in stream order).
bool fs_pointwise_scatter_eligible(int Cin, int Cout, int R, int S, int stride, int dil) {
  return stride >= R && stride >= S && stride > 1 && dil == 1 && Cout % 64 == 0 && Cin % 4 == 0 && Cin >= 32 && R * S <= 9;
}

int fs_pointwise_scatter_conv(int mode, const float* dy, const float* w, float* dx, void* ws, const unsigned* w_amax, int B, int H, int W,
                              int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, hipStream_t stream) {
  if (!fs_pointwise_scatter_eligible(Cin, Cout, R, S, stride, 1)) return FS_ERR_ARG;
  const long pack_bytes = fs_pointwise_pack_bytes(mode, Cout, Cin);
  if (pack_bytes >= 2147483647L || (size_t)B * H * W * Cin * 4 >= 4294967000UL || (size_t)B * Ho * Wo * Cout * 4 >= 4294967000UL) return FS_ERR_ARG;
  for (int r = 0; r < R; ++r)
    for (int s_ = 0; s_ < S; ++s_) {
      // rows: every (oy, ox) whose pixel (oy * st + r - pad, ox * st + s - pad) can lie inside dX, also past the last real dY row / column
      const int dyo = r - pad, dxo = s_ - pad;
      const int He = dyo > H - 1 ? 0 : (H - 1 - dyo) / stride + 1, We = dxo > W - 1 ? 0 : (W - 1 - dxo) / stride + 1;
      if (He <= 0 || We <= 0) continue;
      PwArgs a;
      a.src = dy; a.ws = reinterpret_cast<const unsigned char*>(ws); a.bias = nullptr; a.dst = dx; a.stats = nullptr;
      a.bn_y = nullptr; a.bn_mask = nullptr; a.bn_mean = nullptr; a.bn_invstd = nullptr; a.add_src = nullptr; a.add_mask = nullptr;
      a.dp_scale = 1.f; a.dp_thresh = 0u; a.dp_key = 0u; a.dp_rows = 0;
  a.dp_scale = 1.f; a.dp_thresh = 0u; a.dp_key = 0u; a.dp_rows = 0;
      a.M = (long)B * He * We; a.Cs = Cout; a.Cd = Cin;
      const int nwp = pw_nw(Cin);
      a.Npad = ((Cin + 64 * nwp - 1) / (64 * nwp)) * 64 * nwp;
      a.nchunk = Cout / 32;
      a.nx = cdiv(a.M, ROWS);
      a.ny = nwp == 2 ? a.Npad / 128 : (Cin + 63) / 64;
      a.src_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * 4);
      a.dst_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
      a.ws_bytes = (unsigned)pack_bytes;
      a.drop_scale = 1.f; a.drop_thresh = 0u; a.drop_key = 0u;
      // the A rows: the (Ho x Wo) dY image read on the extended (He x We) grid, "stride 1, pad 0, one tap"
      a.ntap = 1; a.S = 1; a.Hx = Ho; a.Wx = Wo; a.Ho = He; a.Wo = We; a.st = 1; a.pad = 0; a.Cin = Cout; a.rounds_per_tap = Cout / (32 * RC);
      a.sc_on = 1; a.sc_st = stride; a.sc_dy = dyo; a.sc_dx = dxo; a.sc_H = H; a.sc_W = W;
      const float* wt = w + (long)(r * S + s_) * Cin * Cout;
      const int e = mode == 2 ? run_pointwise<PrecF16>(a, wt, ws, w_amax, Cin, Cout, 1, nwp, stream)
                              : run_pointwise<PrecX3>(a, wt, ws, w_amax, Cin, Cout, 1, nwp, stream);
      if (e != FS_OK) return e;
    }
  return FS_OK;
}
