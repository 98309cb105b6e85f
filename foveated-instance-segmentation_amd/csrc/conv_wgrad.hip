// Weight gradient of a convolution on the 16-bit MFMA pipe in split precision (conv_split.h: f16x2 or bf16x3), one launch
// per tap class (conv_kernels.h FsTapClass; 3x3 stride 1 = one class of 9 taps):
//     dW[r][s][ci][co] = sum_{b,y,x} X[b][sm*(y+tr)+cy][sm*(x+ts)+cx][ci] * dY[b][y][x][co],   (r, s) = class tap (tr, ts)
//
// GEMM view: M = ci, N = co, K = pixels.  A workgroup owns one 64(ci) x 64(co) tile of ALL taps of the class (NR*NS 32x32
// accumulators per wave) and a contiguous range of pixel patches (split-K over patches, fp32 atomics at the end).
// Per patch (Ph x Pw <= 64 pixels of one image) it loads the dY patch and the (Ph+NR-1) x (Pw+NS-1) X halo once, (scales and)
// splits both into 16-bit planes while writing them to LDS in their natural [pixel][channel] order, and every tap reads the
// SAME X image at a slot offset of (tr*(Pw+NS-1) + ts).  Both MFMA operands need K (= pixel) contiguous per lane, i.e. the
// transpose of the LDS image: that is what ds_read_b64_tr_b16 delivers (4 pixel rows x 16 channels per 16-lane group,
// column-major), two of them per 8-deep fragment.
//
// LDS: X  [NPL planes][2 channel halves][112 slots][32 ch]  (64-B rows: 4 consecutive slots of one half cover all 64 banks
//      dY [NPL planes][2 channel halves][ 64 pix  ][32 ch]   exactly once)  -> 45 KB (f16x2) / 67.5 KB (bf16x3): two per CU
//
// f16x2 scaling: per patch the workgroup takes max|X| over the halo tile and max|dY| over the patch (exponents ex, ey) and keeps
// a running exponent E of the accumulators' unit (products are accumulated in units of 2^(E-28)).  If ex + ey > E the
// accumulators are rescaled by the exact power of two and E = ex + ey; dY is scaled by 2^(14-ey) and X by 2^(14-(E-ey))
// (<= 2^(14-ex): never overflows; a patch far below the running magnitude loses low-order bits only relative to what is already
// accumulated).  The atomics at the end add acc * 2^(E-28).
#include "conv_split.h"
#include "conv_kernels.h"
#include <stdio.h>
#include <math.h>
#include <stdlib.h>

namespace {

using namespace fs_split;

constexpr int XS = 112;               // halo slots per patch (max)
constexpr int YP = 64;                // pixels per patch (max)
constexpr int X_HALF = XS * 64;       // bytes of one 32-channel half image
constexpr int X_PLANE = 2 * X_HALF;
constexpr int Y_HALF = YP * 64;
constexpr int Y_PLANE = 2 * Y_HALF;
constexpr int NXI = XS * 16 / 256;    // 7 float4 loads per thread for the X halo
constexpr int NYI = YP * 16 / 256;    // 4 for the dY patch
// -DFS_WGRAD_TRACE (kernel A/B builds only; tools/wgrad_trace.sh): time stamps of every wave of the 3x3 class kernel around the phases of
// its patch rounds 2..5 -- loads issued | barrier | loads landed | split + LDS stores | barrier | MFMA loop -- averaged per launch on stderr.
#ifdef FS_WGRAD_TRACE
#include <cstdio>
#define WG_TRACE_FIELD long long* dbg;
#define WG_STAMP(i) do { if (a.dbg != nullptr && lane == 0 && it >= 2 && it < 6) a.dbg[(((long)blockIdx.x * 4 + wave) * 4 + (it - 2)) * 8 + (i)] = clock64(); } while (0)
#define WG_WAIT_LOADS() asm volatile("s_waitcnt vmcnt(0)" ::: "memory")
#else
#define WG_TRACE_FIELD
#define WG_STAMP(i) do { } while (0)
#define WG_WAIT_LOADS() do { } while (0)
#endif
struct WgArgs {
  WG_TRACE_FIELD
  const float* x;    // (B,Hx,Wx,Cin)
  const float* dy;   // (B,H,W,Cout)
  float* dw;         // [R][S][Cin][Cout], zero-initialised or accumulated into
  int B, H, W, Hx, Wx, Cin, Cout;
  // tap class (conv_kernels.h FsTapClass): X row of (dY row oy, class tap tr) = sm*(oy + tr) + cy, filter row rbase + rstep*tr
  int sm, cy, cx, rbase, rstep, sbase, sstep, S;
  int Ph, Pw, tiles_y, tiles_x, npatch, patches_per_split;
  int tiles_ci, tiles_co;
  unsigned x_bytes, dy_bytes;
  unsigned magic_wh, magic_pw;      // 2^32 / (Pw + NS - 1) + 1, 2^32 / Pw + 1: slot / pixel index -> (row, column) without integer division
  int prio;                         // 1: issue priority rises through the MFMA loop (see the kernel)
  FsPart part;                      // deterministic mode: slab `split` takes this workgroup's tile (conv_kernels.h)
};

// 256 threads, two workgroups per CU.  (Measured and rejected, profiles/r02/wgrad_groups_*.txt: a 512-thread workgroup of two patch
// streams that add their accumulators through LDS before the atomics -- half the atomic traffic, 7 % faster alone in f16x2, +-0 in
// bf16x3, 0.5-1.5 % slower in the training step where other streams' workgroups no longer fit beside it.)
template <class P, int NR, int NS>
__global__ __launch_bounds__(256, 2) void conv_wgrad_class_kernel(WgArgs a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NT = NR * NS, NPL = P::NPL;
  constexpr int LDS_BYTES = NPL * (X_PLANE + Y_PLANE);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ unsigned amax_cell[2][2];      // [patch parity][X, dY]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // the operand images start at byte gx (X planes) / gy (dY planes) of `lds`; folded into the precomputed offsets below
  constexpr int gx = 0, gy = NPL * X_PLANE;
  unsigned char* const Xl = lds;
  unsigned char* const Yl = lds;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = a.tiles_ci * a.tiles_co;
  // XCD-aware order: workgroups are dealt round-robin over the 8 XCDs; give each XCD a contiguous range so the
  // ntile workgroups of one split (same pixels, different channel tiles) share one L2.
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rmd = nwg & 7;
  const int wg = (xcd < rmd ? xcd * (qd + 1) : rmd * (qd + 1) + (xcd - rmd) * qd) + loc;
  const int split = wg / ntile, tile = wg - split * ntile;
  const int tci = tile / a.tiles_co, tco = tile - tci * a.tiles_co;
  const int ci0 = tci * 64, co0 = tco * 64;
  const int p_begin = split * a.patches_per_split;
  const int p_end = (p_begin + a.patches_per_split < a.npatch) ? p_begin + a.patches_per_split : a.npatch;
  const int Wh = a.Pw + NS - 1, nslots = (a.Ph + NR - 1) * Wh, npix = a.Ph * a.Pw, nk = (npix + 15) >> 4;
  const int tpi = a.tiles_y * a.tiles_x;

  // ---- loader constants: item i of a thread = (row (tid>>4) + 16 i, channel quad cq) ----
  const int cq = tid & 15;
  int xcode[NXI], ycode[NYI];     // (hy << 16 | hx) of the halo slot / (py << 16 | px) of the patch pixel, -1 = unused
#pragma unroll
  for (int i = 0; i < NXI; ++i) {
    const int slot = (tid >> 4) + 16 * i;
    const int hy = div_small1(slot, a.magic_wh), hx = slot - hy * Wh;
    xcode[i] = (slot < nslots && ci0 + 4 * cq < a.Cin) ? ((hy << 16) | hx) : -1;
  }
#pragma unroll
  for (int i = 0; i < NYI; ++i) {
    const int p = (tid >> 4) + 16 * i;
    const int py = div_small1(p, a.magic_pw), px = p - py * a.Pw;
    ycode[i] = (p < npix && co0 + 4 * cq < a.Cout) ? ((py << 16) | px) : -1;
  }
  // byte offset of item i relative to the patch origin (the patch origin itself is wave-uniform: scalar arithmetic per patch, no
  // per-item integer multiplies -- v_mul_lo_u32 is a quarter-rate instruction and there were 25 of them per patch)
  int xdelta[NXI], ydelta[NYI];
#pragma unroll
  for (int i = 0; i < NXI; ++i) xdelta[i] = ((a.sm * (xcode[i] >> 16) * a.Wx + a.sm * (xcode[i] & 0xffff)) * a.Cin + ci0 + 4 * cq) * 4;
#pragma unroll
  for (int i = 0; i < NYI; ++i) ydelta[i] = (((ycode[i] >> 16) * a.W + (ycode[i] & 0xffff)) * a.Cout + co0 + 4 * cq) * 4;
  const int xw = gx + (cq >> 3) * X_HALF + (tid >> 4) * 64 + (cq & 7) * 8;     // + i*1024 + plane*X_PLANE
  const int yw = gy + (cq >> 3) * Y_HALF + (tid >> 4) * 64 + (cq & 7) * 8;     // + i*1024 + plane*Y_PLANE
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dy, a.dy_bytes);

  // ---- transposed-read lane constants: 16-lane group g reads rows (pixels) q = 0..3, columns cb + 4pp .. +3 ----
  const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g = lane >> 4, lh = g >> 1, cb = 16 * (g & 1);
  int xb[4][2];
#pragma unroll
  for (int ks = 0; ks < 4; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int pidx = 16 * ks + 8 * lh + 4 * t + q;
      if (pidx >= npix) pidx = 0;             // padded k: dY is zero there, any valid X address will do
      const int py = div_small1(pidx, a.magic_pw), px = pidx - py * a.Pw;
      xb[ks][t] = gx + wm * X_HALF + (py * Wh + px) * 64 + (cb + 4 * pp) * 2;
    }
  const int yb = gy + wn * Y_HALF + (8 * lh + q) * 64 + (cb + 4 * pp) * 2;      // + ks*1024 + t*256 + plane*Y_PLANE
  const int rowoff1 = Wh * 64;

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto tr = [&](const unsigned char* base, int off) -> X4 { return P::tr_read(base + off); };
  auto cat = [](X4 lo, X4 hi) -> X8 { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); };
  if (tid < 4) amax_cell[tid >> 1][tid & 1] = 0u;
  __syncthreads();
  int E = 2 * EMIN - 1, par = 0;

  const int niter = p_end - p_begin;
  // (image, tile row, tile column) of the current patch, advanced by one per round (wave-uniform: scalar adds and compares instead
  // of two integer divisions per patch)
  int pb_, pty, ptx;
  {
    pb_ = p_begin / tpi;
    const int trem = p_begin - pb_ * tpi;
    pty = trem / a.tiles_x; ptx = trem - pty * a.tiles_x;
  }
  for (int it = 0; it < niter; ++it) {
    constexpr bool pvalid = true;
    const int b = pb_, ty = pty, tx = ptx;
    const int y0 = ty * a.Ph, x0 = tx * a.Pw;
    if (++ptx >= a.tiles_x) { ptx = 0; if (++pty >= a.tiles_y) { pty = 0; ++pb_; } }

    f32x4 rx[NXI], ry[NYI];
    WG_STAMP(0);
    const int iy0 = a.sm * y0 + a.cy, ix0 = a.sm * x0 + a.cx;                       // source pixel of halo slot (0, 0)
    const int xbase = ((b * a.Hx + iy0) * a.Wx + ix0) * a.Cin * 4;                  // may be negative (padding); base + delta is not, where valid
    const int ybase = ((b * a.H + y0) * a.W + x0) * a.Cout * 4;
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      const int iy = iy0 + a.sm * (xcode[i] >> 16), ix = ix0 + a.sm * (xcode[i] & 0xffff);
      const bool ok = pvalid && xcode[i] >= 0 && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? xbase + xdelta[i] : (int)OOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      const int y = y0 + (ycode[i] >> 16), x = x0 + (ycode[i] & 0xffff);
      const bool ok = pvalid && ycode[i] >= 0 && y < a.H && x < a.W;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, ok ? ybase + ydelta[i] : (int)OOB, 0, 0));
    }
    WG_STAMP(1);
    float sx = 1.f, sy = 1.f;
    if (P::SCALED) {   // tile maxima -> LDS cells of this patch's parity
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int i = 0; i < NXI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(rx[i][e]));
#pragma unroll
      for (int i = 0; i < NYI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) my = fmaxf(my, fabsf(ry[i][e]));
      mx = wave_max(mx); my = wave_max(my);
      if (lane == 0) { atomicMax(&amax_cell[par][0], __builtin_bit_cast(unsigned, mx)); atomicMax(&amax_cell[par][1], __builtin_bit_cast(unsigned, my)); }
    }
    __syncthreads();     // maxima complete; every wave has finished reading the previous patch
    WG_STAMP(2);
    WG_WAIT_LOADS();
    WG_STAMP(3);
    if (P::SCALED) {
      const int ex = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par][0]));
      const int ey = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par][1]));
      if (ex + ey > E) {
        const float f = pow2f(E - ex - ey);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] *= f;
        E = ex + ey;
      }
      sx = pow2f(14 - (E - ey)); sy = pow2f(14 - ey);
      par ^= 1;
      if (tid < 2) amax_cell[par][tid] = 0u;
    }
#pragma unroll
    for (int i = 0; i < NXI; ++i) {
      X4 p[NPL];
      P::split4(P::SCALED ? rx[i] * sx : rx[i], p);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Xl[xw + i * 1024 + pl * X_PLANE]) = p[pl];
    }
#pragma unroll
    for (int i = 0; i < NYI; ++i) {
      X4 p[NPL];
      P::split4(P::SCALED ? ry[i] * sy : ry[i], p);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Yl[yw + i * 1024 + pl * Y_PLANE]) = p[pl];
    }
    WG_STAMP(4);
    __syncthreads();
    WG_STAMP(5);

#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      if (ks < nk) {
        // a.prio (A/B builds): issue priority rises through the MFMA loop and drops to 0 for the load / split phase.  The SIMD arbitrates
        // by priority, then AGE: at equal priority the older of the two resident waves takes the matrix pipe whenever it wants it and
        // the younger workgroup only progresses in the older one's load / split windows (FS_WGRAD_TRACE: MFMA loop 8 500 cycles in wave
        // slot 0, 18 400 in slot 1; once the older workgroup has left, the younger runs alone with the pipe idle during its own
        // load / split).  With the priority growing, the wave that entered its MFMA loop first finishes it at full rate and the two
        // workgroups alternate (both 9 700) -- see launch_class for why that is not the default.
        if (a.prio) {
          if (ks == 0) __builtin_amdgcn_s_setprio(1);
          else if (ks == 1) __builtin_amdgcn_s_setprio(2);
          else if (ks == 2) __builtin_amdgcn_s_setprio(3);
        }
        X8 fb[NPL], fa[2][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fb[pl] = cat(tr(Yl, yb + ks * 1024 + pl * Y_PLANE), tr(Yl, yb + ks * 1024 + 256 + pl * Y_PLANE));
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fa[0][pl] = cat(tr(Xl, xb[ks][0] + pl * X_PLANE), tr(Xl, xb[ks][1] + pl * X_PLANE));
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
          if (tap + 1 < NT) {
            const int r = (tap + 1) / NS, s = (tap + 1) - NS * r;
            const int ro = r * rowoff1 + s * 64;
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl)
              fa[(tap + 1) & 1][pl] = cat(tr(Xl, xb[ks][0] + ro + pl * X_PLANE), tr(Xl, xb[ks][1] + ro + pl * X_PLANE));
          }
          __builtin_amdgcn_sched_barrier(0);
          const X8(&A)[NPL] = fa[tap & 1];
#pragma unroll
          for (int t = 0; t < P::NTERM; ++t) acc[tap] = P::mfma(A[P::ta(t)], fb[P::tb(t)], acc[tap]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
    if (a.prio) __builtin_amdgcn_s_setprio(0);
    WG_STAMP(6);
#ifdef FS_WGRAD_TRACE
    if (a.dbg != nullptr && lane == 0 && it == 2) a.dbg[(((long)blockIdx.x * 4 + wave) * 4) * 8 + 7] = (long long)__builtin_amdgcn_s_getreg((31 << 11) | 4);
#endif
  }

  const int co = co0 + 32 * wn + (lane & 31);
  const int Eo = E - 28;                                   // two factors: the combined exponent can leave the float range
  const float fo1 = P::SCALED ? pow2f(Eo / 2) : 1.f, fo2 = P::SCALED ? pow2f(Eo - Eo / 2) : 1.f;
  if (co < a.Cout && p_begin < p_end) {
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
      const int tr_ = tap / NS, ts_ = tap - NS * tr_;
      const long ftap = (long)(a.rbase + a.rstep * tr_) * a.S + (a.sbase + a.sstep * ts_);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[tap][r] * fo1 * fo2;
        if (ci < a.Cin) fs_wgrad_out(a.dw, a.part, split, (ftap * a.Cin + ci) * a.Cout + co, v);
      }
    }
  }
}

// ---- 3x3 filters with stride 2 or 3: all nine taps in ONE launch (round 4) -----------------------------------------------------------
// One launch per tap class (above) loads and splits the same dY patch once per class -- four times for a 3x3 stride-2 layer -- and its
// 1- and 2-tap classes run 24 / 48 MFMAs per wave between two barriers (SQ counters, profiles/r04/pmc: matrix pipe 13 % busy, 21 VALU
// instructions per MFMA on the single-tap class).  Here a workgroup keeps nine accumulators like the stride-1 kernel and the X halo of
// its dY patch lives in LDS as PARITY PLANES: tap (r, s) reads source pixel (st*py + r, st*px + s) relative to the halo origin, i.e.
// plane (r mod st, s mod st) at in-plane position (py + r div st, px + s div st) -- dense in (py, px), so the transposed fragment reads
// are the conflict-free ones of the class kernel, at a per-tap constant offset.  Every X pixel and every dY pixel is loaded and split
// once.  Patch = Ph x Pw <= 32 dY pixels (two k-steps), <= 160 halo slots over all planes: 73.7 KB of LDS in bf16x3, two per CU.
constexpr int MP_XS = 160;                 // X slots over all planes
constexpr int MP_YP = 32;                  // dY pixels per patch
constexpr int MP_XH = MP_XS * 64, MP_XPL = 2 * MP_XH;
constexpr int MP_YH = MP_YP * 64, MP_YPL = 2 * MP_YH;
constexpr int MP_NXI = MP_XS * 16 / 256;   // 10 float4 loads per thread for the X halo
constexpr int MP_NYI = MP_YP * 16 / 256;   // 2 for the dY patch
struct MpArgs {
  const float* x; const float* dy; float* dw;
  int B, H, W, Hx, Wx, Cin, Cout;          // dY (B,H,W,Cout), X (B,Hx,Wx,Cin)
  int st, pad;
  int Ph, Pw, tiles_y, tiles_x, npatch, patches_per_split, tiles_ci, tiles_co;
  int nplanes, nslots;
  int pl_base[9], pl_wcol[9], pl_pr[9], pl_pc[9];      // plane: first slot, row width, residue of its source rows / columns
  unsigned pl_magic[9];                                // div_small1 magic of the row width
  int tap_off[9];                                      // LDS byte offset of tap (r, s): (base + (r div st) * wcol + s div st) * 64
  int tap_w[9];                                        // 1: the tap's plane has row width Pw + 1, 0: Pw
  unsigned x_bytes, dy_bytes, magic_pw;
  FsPart part;
};

template <class P>
__global__ __launch_bounds__(256, 2) void conv_wgrad_planes_kernel(MpArgs a) {
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NT = 9, NPL = P::NPL;
  constexpr int LDS_BYTES = NPL * (MP_XPL + MP_YPL);
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  __shared__ unsigned amax_cell[2][2];      // [patch parity][X, dY]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  constexpr int gx = 0, gy = NPL * MP_XPL;
  unsigned char* const Xl = lds;
  unsigned char* const Yl = lds;
  const int wm = wave >> 1, wn = wave & 1;
  const int ntile = a.tiles_ci * a.tiles_co;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rmd = nwg & 7;
  const int wg = (xcd < rmd ? xcd * (qd + 1) : rmd * (qd + 1) + (xcd - rmd) * qd) + loc;
  const int split = wg / ntile, tile = wg - split * ntile;
  const int tci = tile / a.tiles_co, tco = tile - tci * a.tiles_co;
  const int ci0 = tci * 64, co0 = tco * 64;
  const int p_begin = split * a.patches_per_split;
  const int p_end = (p_begin + a.patches_per_split < a.npatch) ? p_begin + a.patches_per_split : a.npatch;
  const int npix = a.Ph * a.Pw, nk = (npix + 15) >> 4;
  const int tpi = a.tiles_y * a.tiles_x;

  // ---- loader constants: item i of a thread = (slot (tid >> 4) + 16 i, channel quad cq); the slot's plane by a compare chain over the
  // (wave-uniform) plane table, once per kernel ----
  const int cq = tid & 15;
  int xcode[MP_NXI];                     // (source row << 16 | source column) relative to the halo origin, -1 = unused (its byte offset is formed per patch: ten fewer registers live across the MFMA loop, which is at the 256-register limit)
#pragma unroll
  for (int i = 0; i < MP_NXI; ++i) {
    const int slot = (tid >> 4) + 16 * i;
    int base = 0, wc = a.pl_wcol[0], pr = a.pl_pr[0], pc = a.pl_pc[0];
    unsigned mg = a.pl_magic[0];
#pragma unroll
    for (int k = 1; k < 9; ++k)
      if (k < a.nplanes && slot >= a.pl_base[k]) { base = a.pl_base[k]; wc = a.pl_wcol[k]; pr = a.pl_pr[k]; pc = a.pl_pc[k]; mg = a.pl_magic[k]; }
    const int local = slot - base;
    const int hy = div_small1(local, mg), hx = local - hy * wc;
    const int ry = a.st * hy + pr, rx = a.st * hx + pc;
    xcode[i] = (slot < a.nslots && ci0 + 4 * cq < a.Cin) ? ((ry << 16) | rx) : -1;
  }
  const int xchan = (ci0 + 4 * cq) * 4;
  int ycode[MP_NYI], ydelta[MP_NYI];
#pragma unroll
  for (int i = 0; i < MP_NYI; ++i) {
    const int p = (tid >> 4) + 16 * i;
    const int py = div_small1(p, a.magic_pw), px = p - py * a.Pw;
    ycode[i] = (p < npix && co0 + 4 * cq < a.Cout) ? ((py << 16) | px) : -1;
    ydelta[i] = ((py * a.W + px) * a.Cout + co0 + 4 * cq) * 4;
  }
  const int xw = gx + (cq >> 3) * MP_XH + (tid >> 4) * 64 + (cq & 7) * 8;     // + i*1024 + plane*MP_XPL
  const int yw = gy + (cq >> 3) * MP_YH + (tid >> 4) * 64 + (cq & 7) * 8;     // + i*1024 + plane*MP_YPL
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dy, a.dy_bytes);

  // ---- transposed-read lane constants (class kernel): 16-lane group g reads k rows q = 0..3, columns cb + 4pp .. +3 ----
  const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g = lane >> 4, lh = g >> 1, cb = 16 * (g & 1);
  int xb[2][2][2];                 // [row width Pw / Pw + 1][k-step][half]: byte offset of the lane's patch pixel in a plane of that width
#pragma unroll
  for (int w = 0; w < 2; ++w)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int t = 0; t < 2; ++t) {
        int pidx = 16 * ks + 8 * lh + 4 * t + q;
        if (pidx >= npix) pidx = 0;             // padded k: dY is zero there, any valid X address will do
        const int py = div_small1(pidx, a.magic_pw), px = pidx - py * a.Pw;
        xb[w][ks][t] = gx + wm * MP_XH + (py * (a.Pw + w) + px) * 64 + (cb + 4 * pp) * 2;
      }
  const int yb = gy + wn * MP_YH + (8 * lh + q) * 64 + (cb + 4 * pp) * 2;      // + ks*1024 + t*256 + plane*MP_YPL

  f32x16 acc[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

  auto tr = [&](const unsigned char* base, int off) -> X4 { return P::tr_read(base + off); };
  auto cat = [](X4 lo, X4 hi) -> X8 { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); };
  if (tid < 4) amax_cell[tid >> 1][tid & 1] = 0u;
  __syncthreads();
  int E = 2 * EMIN - 1, par = 0;

  const int niter = p_end - p_begin;
  int pb_, pty, ptx;
  {
    pb_ = p_begin / tpi;
    const int trem = p_begin - pb_ * tpi;
    pty = trem / a.tiles_x; ptx = trem - pty * a.tiles_x;
  }
  for (int it = 0; it < niter; ++it) {
    const int b = pb_, ty = pty, tx = ptx;
    const int y0 = ty * a.Ph, x0 = tx * a.Pw;
    if (++ptx >= a.tiles_x) { ptx = 0; if (++pty >= a.tiles_y) { pty = 0; ++pb_; } }

    f32x4 rx[MP_NXI], ry[MP_NYI];
    const int iy0 = a.st * y0 - a.pad, ix0 = a.st * x0 - a.pad;                    // source pixel of the halo origin
    const int xbase = ((b * a.Hx + iy0) * a.Wx + ix0) * a.Cin * 4;                  // may be negative (padding); base + delta is not, where valid
    const int ybase = ((b * a.H + y0) * a.W + x0) * a.Cout * 4;
#pragma unroll
    for (int i = 0; i < MP_NXI; ++i) {
      const int ry_ = xcode[i] >> 16, rx_ = xcode[i] & 0xffff;
      const int iy = iy0 + ry_, ix = ix0 + rx_;
      const bool ok = xcode[i] >= 0 && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? xbase + (ry_ * a.Wx + rx_) * a.Cin * 4 + xchan : (int)OOB, 0, 0));
    }
#pragma unroll
    for (int i = 0; i < MP_NYI; ++i) {
      const int y = y0 + (ycode[i] >> 16), x = x0 + (ycode[i] & 0xffff);
      const bool ok = ycode[i] >= 0 && y < a.H && x < a.W;
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, ok ? ybase + ydelta[i] : (int)OOB, 0, 0));
    }
    float sx = 1.f, sy = 1.f;
    if (P::SCALED) {   // tile maxima -> LDS cells of this patch's parity (running exponents as in the class kernel)
      float mx = 0.f, my = 0.f;
#pragma unroll
      for (int i = 0; i < MP_NXI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) mx = fmaxf(mx, fabsf(rx[i][e]));
#pragma unroll
      for (int i = 0; i < MP_NYI; ++i)
#pragma unroll
        for (int e = 0; e < 4; ++e) my = fmaxf(my, fabsf(ry[i][e]));
      mx = wave_max(mx); my = wave_max(my);
      if (lane == 0) { atomicMax(&amax_cell[par][0], __builtin_bit_cast(unsigned, mx)); atomicMax(&amax_cell[par][1], __builtin_bit_cast(unsigned, my)); }
    }
    __syncthreads();     // maxima complete; every wave has finished reading the previous patch
    if (P::SCALED) {
      const int ex = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par][0]));
      const int ey = __builtin_amdgcn_readfirstlane(exponent_of_bits(amax_cell[par][1]));
      if (ex + ey > E) {
        const float f = pow2f(E - ex - ey);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
          for (int r = 0; r < 16; ++r) acc[t][r] *= f;
        E = ex + ey;
      }
      sx = pow2f(14 - (E - ey)); sy = pow2f(14 - ey);
      par ^= 1;
      if (tid < 2) amax_cell[par][tid] = 0u;
    }
#pragma unroll
    for (int i = 0; i < MP_NXI; ++i) {
      X4 p[NPL];
      P::split4(P::SCALED ? rx[i] * sx : rx[i], p);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Xl[xw + i * 1024 + pl * MP_XPL]) = p[pl];
    }
#pragma unroll
    for (int i = 0; i < MP_NYI; ++i) {
      X4 p[NPL];
      P::split4(P::SCALED ? ry[i] * sy : ry[i], p);
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Yl[yw + i * 1024 + pl * MP_YPL]) = p[pl];
    }
    __syncthreads();

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < nk) {
        X8 fb[NPL], fa[2][NPL];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) fb[pl] = cat(tr(Yl, yb + ks * 1024 + pl * MP_YPL), tr(Yl, yb + ks * 1024 + 256 + pl * MP_YPL));
        {
          const int o0 = (a.tap_w[0] ? xb[1][ks][0] : xb[0][ks][0]) + a.tap_off[0], o1 = (a.tap_w[0] ? xb[1][ks][1] : xb[0][ks][1]) + a.tap_off[0];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl) fa[0][pl] = cat(tr(Xl, o0 + pl * MP_XPL), tr(Xl, o1 + pl * MP_XPL));
        }
#pragma unroll
        for (int tap = 0; tap < NT; ++tap) {
          if (tap + 1 < NT) {
            const int tn = tap + 1;
            const int o0 = (a.tap_w[tn] ? xb[1][ks][0] : xb[0][ks][0]) + a.tap_off[tn], o1 = (a.tap_w[tn] ? xb[1][ks][1] : xb[0][ks][1]) + a.tap_off[tn];
#pragma unroll
            for (int pl = 0; pl < NPL; ++pl) fa[tn & 1][pl] = cat(tr(Xl, o0 + pl * MP_XPL), tr(Xl, o1 + pl * MP_XPL));
          }
          __builtin_amdgcn_sched_barrier(0);
          const X8(&A)[NPL] = fa[tap & 1];
#pragma unroll
          for (int t = 0; t < P::NTERM; ++t) acc[tap] = P::mfma(A[P::ta(t)], fb[P::tb(t)], acc[tap]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
    }
  }

  const int co = co0 + 32 * wn + (lane & 31);
  const int Eo = E - 28;
  const float fo1 = P::SCALED ? pow2f(Eo / 2) : 1.f, fo2 = P::SCALED ? pow2f(Eo - Eo / 2) : 1.f;
  if (co < a.Cout && p_begin < p_end) {
#pragma unroll
    for (int tap = 0; tap < NT; ++tap) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        const float v = acc[tap][r] * fo1 * fo2;
        if (ci < a.Cin) fs_wgrad_out(a.dw, a.part, split, ((long)tap * a.Cin + ci) * a.Cout + co, v);
      }
    }
  }
}

// ---- 3x3 / stride 1 / pad 1 weight gradient in the F(2,3) transform domain (bf16x3, even widths; round 3) -------------------------
// The conv kernels are power-limited, so what shortens them is issuing fewer MFMAs.  The forward's row transform
//     out[2j] = m0 + m1 + m2,  out[2j+1] = m1 - m2 - m3,   m_c = sum_{ky, ci} T_c(x row y+ky-1, pair j) * U_c(g row ky)
// gives, for the loss gradient e0 = dY[2j], e1 = dY[2j+1] of a PAIR of output pixels,
//     dU_c[ky][ci][co] = sum over pairs of T_c[ci] * dM_c[co],     dM = (e0, e0 + e1, e0 - e1, -e1),
//     dW[ky][0] = dU0 + (dU1 + dU2)/2,   dW[ky][1] = (dU1 - dU2)/2,   dW[ky][2] = (dU1 + dU2)/2 + dU3      (U = G g transposed)
// i.e. 12 accumulator tiles over K = pixel PAIRS instead of 9 over K = pixels: 2/3 of the MFMA work.  T and dM are formed in fp32 and
// then split into the three bf16 planes (both operands keep 24 bits), as in conv_wino.hip.
//   workgroup = 512 threads = 8 waves = 4 quadrants of the 64 (ci) x 64 (co) tile x 2 component pairs {0,1} / {2,3}; a wave holds
//   3 filter rows x 2 components = 6 accumulators.  Patch = Ph rows x PP pairs <= 32 pairs, halo (Ph + 2) x PP <= 48 pair slots.
//   LDS: T  [3 planes][4 comps][2 channel halves][48 slots][32 ch] 73.7 KB + dM [3][4][2][32 pairs][32 ch] 49.2 KB: one workgroup per CU.
//   Operand fragments by ds_read_b64_tr_b16 exactly as in conv_wgrad_class_kernel (K = pair index contiguous per lane).
constexpr int WP_SLOTS = 48, WP_PAIRS = 32;
constexpr int XT_HALF = WP_SLOTS * 64, XT_COMP = 2 * XT_HALF, XT_PLANE = 4 * XT_COMP;      // bytes
constexpr int DM_HALF = WP_PAIRS * 64, DM_COMP = 2 * DM_HALF, DM_PLANE = 4 * DM_COMP;
struct WwArgs {
  const float* x; const float* dy; float* dw;
  int B, H, W, Cin, Cout;
  int Ph, PP, tiles_y, tiles_x, npatch, patches_per_split, tiles_ci, tiles_co;
  unsigned x_bytes, dy_bytes, magic_pp;
  FsPart part;                      // deterministic mode: slab 2 * split + cp (the two component-pair waves add to the same elements)
};

template <class P>
__global__ __launch_bounds__(512, 1) void conv_wgrad_wino_kernel(WwArgs a) {
  static_assert(!P::SCALED, "bf16x3 only");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];       // T image, then dM image
  constexpr int DM0 = NPL * XT_PLANE;

  const int tid = threadIdx.x, lane = tid & 63;
  const int hw = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int cp = hw >> 2, wm = (hw >> 1) & 1, wn = hw & 1;
  const int ntile = a.tiles_ci * a.tiles_co;
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rmd = nwg & 7;
  const int wg = (xcd < rmd ? xcd * (qd + 1) : rmd * (qd + 1) + (xcd - rmd) * qd) + loc;
  const int split = wg / ntile, tile = wg - split * ntile;
  const int tci = tile / a.tiles_co, tco = tile - tci * a.tiles_co;
  const int ci0 = tci * 64, co0 = tco * 64;
  const int p_begin = split * a.patches_per_split;
  const int p_end = (p_begin + a.patches_per_split < a.npatch) ? p_begin + a.patches_per_split : a.npatch;
  const int nslots = (a.Ph + 2) * a.PP, npairs = a.Ph * a.PP, nk = (npairs + 15) >> 4;
  const int tpi = a.tiles_y * a.tiles_x;

  // ---- loaders: X item i = (halo slot (tid >> 4) + 32 i, channel quad tid & 15), 4 pixels d0..d3 each; dY item = (pair tid >> 4, quad) ----
  const int cq = tid & 15;
  int xcode[2], xdelta[2];
#pragma unroll
  for (int i = 0; i < 2; ++i) {
    const int slot = (tid >> 4) + 32 * i;
    const int hy = div_small1(slot, a.magic_pp), pj = slot - hy * a.PP;
    xcode[i] = (slot < nslots && ci0 + 4 * cq < a.Cin) ? ((hy << 16) | pj) : -1;
    xdelta[i] = ((hy * a.W + 2 * pj) * a.Cin + ci0 + 4 * cq) * 4;        // relative to pixel (y0 - 1, x0 - 1)
  }
  int ycode, ydelta;
  {
    const int pr = tid >> 4;
    const int py = div_small1(pr, a.magic_pp), px = pr - py * a.PP;
    ycode = (pr < npairs && co0 + 4 * cq < a.Cout) ? ((py << 16) | px) : -1;
    ydelta = ((py * a.W + 2 * px) * a.Cout + co0 + 4 * cq) * 4;
  }
  const int xw = (cq >> 3) * XT_HALF + (tid >> 4) * 64 + (cq & 7) * 8;        // + i * 32 slots * 64 + comp * XT_COMP + plane * XT_PLANE
  const int yw = DM0 + (cq >> 3) * DM_HALF + (tid >> 4) * 64 + (cq & 7) * 8;
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dy, a.dy_bytes);

  // ---- transposed-read lane constants (conv_wgrad_class_kernel): 16-lane group g reads k rows q = 0..3, columns cb + 4 pp .. + 3 ----
  const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g = lane >> 4, lh = g >> 1, cb = 16 * (g & 1);
  int xb[2][2];
#pragma unroll
  for (int ks = 0; ks < 2; ++ks)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      int pidx = 16 * ks + 8 * lh + 4 * t + q;
      if (pidx >= npairs) pidx = 0;             // padded k: dM is zero there, any valid T address will do
      const int py = div_small1(pidx, a.magic_pp), px = pidx - py * a.PP;
      xb[ks][t] = wm * XT_HALF + (py * a.PP + px) * 64 + (cb + 4 * pp) * 2;
    }
  const int yb = DM0 + wn * DM_HALF + (8 * lh + q) * 64 + (cb + 4 * pp) * 2;      // + ks * 1024 + t * 256 + comp * DM_COMP + plane * DM_PLANE
  const int rowoff = a.PP * 64;                                                   // one halo row down (filter row ky)

  f32x16 acc[3][2];       // [filter row][component of this wave's pair]
#pragma unroll
  for (int ky = 0; ky < 3; ++ky)
#pragma unroll
    for (int cc = 0; cc < 2; ++cc)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[ky][cc][r] = 0.f;

  auto tr = [&](int off) -> X4 { return P::tr_read(lds + off); };
  auto cat = [](X4 lo, X4 hi) -> X8 { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); };

  int pb_, pty, ptx;
  {
    pb_ = p_begin / tpi;
    const int trem = p_begin - pb_ * tpi;
    pty = trem / a.tiles_x; ptx = trem - pty * a.tiles_x;
  }
  f32x4 rx[2][4], ry[2];
  auto load_patch = [&]() {
    const int b = pb_, y0 = pty * a.Ph, x0 = ptx * 2 * a.PP;
    if (++ptx >= a.tiles_x) { ptx = 0; if (++pty >= a.tiles_y) { pty = 0; ++pb_; } }
    const int xbase = ((b * a.H + y0 - 1) * a.W + x0 - 1) * a.Cin * 4;              // may be negative (padding); base + delta is not, where valid
    const int ybase = ((b * a.H + y0) * a.W + x0) * a.Cout * 4;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && hw >= 4) continue;                                               // slots 32..47 belong to waves 0..3 (wave-uniform)
      const int iy = y0 - 1 + (xcode[i] >> 16), ix = x0 - 1 + 2 * (xcode[i] & 0xffff);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const bool ok = xcode[i] >= 0 && iy >= 0 && iy < a.H && ix + e >= 0 && ix + e < a.W;
        rx[i][e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, ok ? xbase + xdelta[i] + e * a.Cin * 4 : (int)OOB, 0, 0));
      }
    }
    const int y = y0 + (ycode >> 16), xx = x0 + 2 * (ycode & 0xffff);
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const bool ok = ycode >= 0 && y < a.H && xx + e < a.W;
      ry[e] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, ok ? ybase + ydelta + e * a.Cout * 4 : (int)OOB, 0, 0));
    }
  };

  const int niter = p_end - p_begin;
  if (niter > 0) load_patch();
  for (int it = 0; it < niter; ++it) {
    __syncthreads();     // every wave has finished reading the previous patch's images
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      if (i == 1 && hw >= 4) continue;
      const f32x4 d0 = rx[i][0], d1 = rx[i][1], d2 = rx[i][2], d3 = rx[i][3];
      const f32x4 T[4] = {d0 - d2, d1 + d2, d2 - d1, d1 - d3};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X4 p[NPL];
        P::split4(T[c], p);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&lds[xw + i * 2048 + c * XT_COMP + pl * XT_PLANE]) = p[pl];
      }
    }
    {
      const f32x4 e0 = ry[0], e1 = ry[1];
      const f32x4 M[4] = {e0, e0 + e1, e0 - e1, -e1};
#pragma unroll
      for (int c = 0; c < 4; ++c) {
        X4 p[NPL];
        P::split4(M[c], p);
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&lds[yw + c * DM_COMP + pl * DM_PLANE]) = p[pl];
      }
    }
    __syncthreads();
    if (it + 1 < niter) load_patch();        // the next patch travels while this one is multiplied

#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      if (ks < nk) {
#pragma unroll
        for (int cc = 0; cc < 2; ++cc) {
          const int c = 2 * cp + cc;
          X8 fb[NPL], fa[2][NPL];
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            fb[pl] = cat(tr(yb + ks * 1024 + c * DM_COMP + pl * DM_PLANE), tr(yb + ks * 1024 + 256 + c * DM_COMP + pl * DM_PLANE));
#pragma unroll
          for (int pl = 0; pl < NPL; ++pl)
            fa[0][pl] = cat(tr(xb[ks][0] + c * XT_COMP + pl * XT_PLANE), tr(xb[ks][1] + c * XT_COMP + pl * XT_PLANE));
#pragma unroll
          for (int ky = 0; ky < 3; ++ky) {
            if (ky + 1 < 3) {
              const int ro = (ky + 1) * rowoff;
#pragma unroll
              for (int pl = 0; pl < NPL; ++pl)
                fa[(ky + 1) & 1][pl] = cat(tr(xb[ks][0] + ro + c * XT_COMP + pl * XT_PLANE), tr(xb[ks][1] + ro + c * XT_COMP + pl * XT_PLANE));
            }
            __builtin_amdgcn_sched_barrier(0);
            const X8(&A)[NPL] = fa[ky & 1];
#pragma unroll
            for (int t = 0; t < P::NTERM; ++t) acc[ky][cc] = P::mfma(A[P::ta(t)], fb[P::tb(t)], acc[ky][cc]);
            __builtin_amdgcn_sched_barrier(0);
          }
        }
      }
    }
  }

  // ---- dU -> dW contributions of this wave's two components, split-K atomics ----
  const int co = co0 + 32 * wn + (lane & 31);
  if (co < a.Cout && p_begin < p_end) {
#pragma unroll
    for (int ky = 0; ky < 3; ++ky)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + 32 * wm + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (ci >= a.Cin) continue;
        const float u0 = acc[ky][0][r], u1 = acc[ky][1][r];
        // cp = 0: (dU0, dU1) -> dW0 += dU0 + dU1/2, dW1 += dU1/2, dW2 += dU1/2;   cp = 1: (dU2, dU3) -> dW0 += dU2/2, dW1 -= dU2/2, dW2 += dU2/2 + dU3
        const float h = cp == 0 ? 0.5f * u1 : 0.5f * u0;
        const float w0 = cp == 0 ? u0 + h : h;
        const float w1 = cp == 0 ? h : -h;
        const float w2 = cp == 0 ? h : h + u1;
        const long d = ((long)(ky * 3) * a.Cin + ci) * a.Cout + co;
        fs_wgrad_out(a.dw, a.part, 2 * split + cp, d, w0);
        fs_wgrad_out(a.dw, a.part, 2 * split + cp, d + (long)a.Cin * a.Cout, w1);
        fs_wgrad_out(a.dw, a.part, 2 * split + cp, d + 2L * a.Cin * a.Cout, w2);
      }
  }
}

// ---- 1x1 / stride 1 weight gradient = the weight gradient of a linear layer: dW[ci][co] = sum_rows X[row][ci] * dY[row][co] (round 3) ----
// The class kernel above pays its per-patch cost (exposed loads, two barriers) for 24 MFMAs per wave when the class is a single tap; here
// a workgroup owns a (WM*MI*32) x (WN*NI*32) tile of dW (128 x 128 for wide layers: 2 x 2 accumulators per wave, half the LDS fragment
// reads per MFMA of a 32 x 32 wave tile) and a contiguous range of rows (split-K, fp32 atomics at the end), walks it in chunks of 32 rows
// (two k-steps of 16), and the next chunk's global loads are in flight during the MFMA phase.  LDS images as in the class kernel:
// [plane][32-channel block][32 rows][32 ch] 16-bit, natural order, both operands read transposed by ds_read_b64_tr_b16.
// Unscaled split only (bf16x3): the f16x2 mode keeps the class kernel and its running exponents.
struct LwArgs {
  const float* x;     // [rows][Cin]
  const float* dy;    // [rows][Cout]
  float* dw;          // [Cin][Cout], zero-initialised or accumulated into
  float* dbias;       // nullable: [Cout] column sums of dy (the layer's bias gradient), zero-initialised or accumulated into
  int rows, Cin, Cout, tiles_ci, tiles_co, rows_per_split;
  unsigned x_bytes, dy_bytes;
  FsPart part;        // deterministic mode: slab `split` of dW images
  float* bpart;       // deterministic mode, with dbias: slab 4 * split + wave of Cout floats (the four waves hold different rows)
  // gathered rows (round 4): the single-tap classes of a convolution whose stride is at least its filter size -- 3x3 stride 4 in the C1
  // classification head, 1x1 stride 2 -- as ntap linear layers in ONE launch.  Row r of dY = output pixel (b, oy, ox); tap (tr, ts) pairs
  // it with X pixel (b, oy * st + tr - pad, ox * st + ts - pad) (zero outside the image) and owns dw + tap * Cin * Cout.  ntap = 0: plain rows.
  int ntap, S, Ho, Wo, Hx, Wx, st, pad;
};

template <class P, int WM, int WN, int MI, int NI>
__global__ __launch_bounds__(256, 2) void linear_wgrad_kernel(LwArgs a) {
  static_assert(WM * WN == 4, "four waves");
  typedef typename P::x8 X8;
  typedef typename P::x4 X4;
  constexpr int NPL = P::NPL;
  constexpr int XB = WM * MI, YB = WN * NI;               // 32-channel blocks of the tile
  constexpr int BLK = 32 * 64;                            // bytes of one block image: 32 rows x 32 channels x 2 B
  constexpr int XPL = XB * BLK, YPL = YB * BLK;
  __shared__ __attribute__((aligned(16))) unsigned char lds[NPL * (XPL + YPL)];
  unsigned char* const Xl = lds;
  unsigned char* const Yl = lds + NPL * XPL;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave / WN, wn = wave - wm * WN;
  const int ntile = a.tiles_ci * a.tiles_co;
  // XCD-aware order (as the class kernel): the ntile workgroups of one split -- same rows, different channel tiles -- share one L2
  const int nwg = gridDim.x;
  const int xcd = blockIdx.x & 7, loc = blockIdx.x >> 3;
  const int qd = nwg >> 3, rmd = nwg & 7;
  const int wg = (xcd < rmd ? xcd * (qd + 1) : rmd * (qd + 1) + (xcd - rmd) * qd) + loc;
  const int per_tap = a.ntap > 0 ? nwg / a.ntap : nwg;           // gathered rows: tap-major, then split, then channel tile
  const int tap = wg / per_tap, wgl = wg - tap * per_tap;
  const int split = wgl / ntile, tile = wgl - split * ntile;
  const int tci = tile / a.tiles_co, tco = tile - tci * a.tiles_co;
  const int ci0 = tci * (XB * 32), co0 = tco * (YB * 32);
  const int k_begin = split * a.rows_per_split;
  const int k_end = k_begin + a.rows_per_split < a.rows ? k_begin + a.rows_per_split : a.rows;

  // loader: item i of a thread = (row tid >> 3 of the chunk, channel quad tid & 7 of block i): 8 lanes read 128 contiguous bytes of a row,
  // a wave writes 8 rows x 64 B = 512 contiguous bytes of one block image
  const int lrow = tid >> 3, quad = tid & 7;
  const __amdgpu_buffer_rsrc_t rsrc_x = make_rsrc(a.x, a.x_bytes);
  const __amdgpu_buffer_rsrc_t rsrc_y = make_rsrc(a.dy, a.dy_bytes);
  int xoff[XB], yoff[YB];        // byte offset of item i in row 0, -1 = channel beyond the tensor
#pragma unroll
  for (int i = 0; i < XB; ++i) { const int c = ci0 + 32 * i + 4 * quad; xoff[i] = c < a.Cin ? c * 4 : -1; }
#pragma unroll
  for (int i = 0; i < YB; ++i) { const int c = co0 + 32 * i + 4 * quad; yoff[i] = c < a.Cout ? c * 4 : -1; }
  f32x4 rx[XB], ry[YB];
  // gathered rows: (image, output row, output column) of this thread's row, advanced by 32 rows per chunk (load_chunk is called once per
  // chunk, in order); the tap's shift of the source pixel
  int gb = 0, goy = 0, gox = 0;
  const int tr_ = a.ntap > 0 ? tap / a.S : 0, ts_ = a.ntap > 0 ? tap - tr_ * a.S : 0;
  if (a.ntap > 0) {
    const int r0 = k_begin + lrow, hw = a.Ho * a.Wo;
    gb = r0 / hw;
    const int rem = r0 - gb * hw;
    goy = rem / a.Wo; gox = rem - goy * a.Wo;
  }
  const long part_tap = (long)tap * a.Cin * a.Cout;
  auto load_chunk = [&](int k) {
    const int r = k + lrow;
    bool rok = r < k_end;
    int xr = r * a.Cin * 4;
    const int yr = r * a.Cout * 4;
    bool xok = rok;
    if (a.ntap > 0) {
      const int iy = goy * a.st + tr_ - a.pad, ix = gox * a.st + ts_ - a.pad;
      xok = rok && iy >= 0 && iy < a.Hx && ix >= 0 && ix < a.Wx;
      xr = ((gb * a.Hx + iy) * a.Wx + ix) * a.Cin * 4;
      gox += 32;
      while (gox >= a.Wo) { gox -= a.Wo; ++goy; }
      while (goy >= a.Ho) { goy -= a.Ho; ++gb; }
    }
#pragma unroll
    for (int i = 0; i < XB; ++i)
      rx[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_x, xok && xoff[i] >= 0 ? xr + xoff[i] : (int)OOB, 0, 0));
#pragma unroll
    for (int i = 0; i < YB; ++i)
      ry[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsrc_y, rok && yoff[i] >= 0 ? yr + yoff[i] : (int)OOB, 0, 0));
  };
  const int wofs = lrow * 64 + quad * 8;     // + block * BLK + plane * XPL / YPL

  // transposed-read lane constants (class kernel): 16-lane group g reads rows q = 0..3 of its 8-row half, columns cb + 4 pp .. + 3
  const int i16 = lane & 15, q = i16 >> 2, pp = i16 & 3, g = lane >> 4, lh = g >> 1, cb = 16 * (g & 1);
  const int rofs = (8 * lh + q) * 64 + (cb + 4 * pp) * 2;      // + ks * 1024 + t * 256 + block * BLK + plane * PL

  f32x16 acc[MI][NI];
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mi][ni][r] = 0.f;

  auto tr = [&](const unsigned char* base, int off) -> X4 { return P::tr_read(base + off); };
  auto cat = [](X4 lo, X4 hi) -> X8 { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); };

  // bias gradient: the workgroups of the first ci tile also sum the dY rows they load anyway (rows beyond the range load as zero)
  const bool want_b = a.dbias != nullptr && tci == 0;
  f32x4 bs[YB];
#pragma unroll
  for (int i = 0; i < YB; ++i) bs[i] = f32x4{0.f, 0.f, 0.f, 0.f};

  if (k_begin < k_end) load_chunk(k_begin);
  for (int k = k_begin; k < k_end; k += 32) {
    if (want_b) {
#pragma unroll
      for (int i = 0; i < YB; ++i) bs[i] += ry[i];
    }
    X4 px[XB][NPL], py[YB][NPL];
#pragma unroll
    for (int i = 0; i < XB; ++i) P::split4(rx[i], px[i]);
#pragma unroll
    for (int i = 0; i < YB; ++i) P::split4(ry[i], py[i]);
    __syncthreads();                 // every wave has finished reading the previous chunk
#pragma unroll
    for (int i = 0; i < XB; ++i)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Xl[wofs + i * BLK + pl * XPL]) = px[i][pl];
#pragma unroll
    for (int i = 0; i < YB; ++i)
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) *reinterpret_cast<X4*>(&Yl[wofs + i * BLK + pl * YPL]) = py[i][pl];
    __syncthreads();
    if (k + 32 < k_end) load_chunk(k + 32);          // in flight during the MFMA phase
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      X8 fa[MI][NPL], fb[NI][NPL];
#pragma unroll
      for (int mi = 0; mi < MI; ++mi)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const int o = rofs + ks * 1024 + (wm * MI + mi) * BLK + pl * XPL;
          fa[mi][pl] = cat(tr(Xl, o), tr(Xl, o + 256));
        }
#pragma unroll
      for (int ni = 0; ni < NI; ++ni)
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const int o = rofs + ks * 1024 + (wn * NI + ni) * BLK + pl * YPL;
          fb[ni][pl] = cat(tr(Yl, o), tr(Yl, o + 256));
        }
#pragma unroll
      for (int t = 0; t < P::NTERM; ++t)          // smallest terms first
#pragma unroll
        for (int mi = 0; mi < MI; ++mi)
#pragma unroll
          for (int ni = 0; ni < NI; ++ni) acc[mi][ni] = P::mfma(fa[mi][P::ta(t)], fb[ni][P::tb(t)], acc[mi][ni]);
    }
  }

  if (k_begin >= k_end) return;
  if (want_b) {        // lanes of a wave with equal channel quad (lane & 7) hold different rows
#pragma unroll
    for (int i = 0; i < YB; ++i)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float v = bs[i][e];
        v += __shfl_xor(v, 8, 64); v += __shfl_xor(v, 16, 64); v += __shfl_xor(v, 32, 64);
        if (lane < 8 && yoff[i] >= 0) {
          if (a.bpart != nullptr) a.bpart[(long)(4 * split + wave) * a.Cout + co0 + 32 * i + 4 * quad + e] = v;
          else atomicAdd(&a.dbias[co0 + 32 * i + 4 * quad + e], v);
        }
      }
  }
#pragma unroll
  for (int mi = 0; mi < MI; ++mi)
#pragma unroll
    for (int ni = 0; ni < NI; ++ni) {
      const int co = co0 + (wn * NI + ni) * 32 + (lane & 31);
      if (co >= a.Cout) continue;
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int ci = ci0 + (wm * MI + mi) * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
        if (ci < a.Cin) fs_wgrad_out(a.dw, a.part, split, part_tap + (long)ci * a.Cout + co, acc[mi][ni][r]);
      }
    }
}

// Split-K count of a small problem.  A launch costs about (K rounds per workgroup) x t_round + (workgroups x tile bytes) / (1 TB/s of fp32
// atomics at the L2: one dword per channel and clock), so the best split count is n* = sqrt(rounds * t_round / (ntile * t_tile)), with
// t_tile the time the L2 needs for one workgroup's tile.  On the HRNet layers n* is at or above the "fill the chip" count used so far
// (64 -> 64 @ 80x80, B = 64: 6 400 patches, one tile: n* = 530), so nothing changes there; on DeepLab's 10x10 maps at 16 images per GPU
// (25 patches, 16 tiles: n* = 8) "fill the chip" launched 400 workgroups that each ran ONE 6 us round and then pushed 147 KB of atomics:
// 81 us per launch for a 1.9 GFLOP problem (gpurun_out/trace_c4, round 4).
static int splitk_model(long rounds, int ntile, double t_round_us, double tile_kb) {
  const double n = sqrt((double)rounds * t_round_us / ((double)ntile * tile_kb * 0.001));      // 1 KB of atomics ~ 1 ns
  return n < 1.0 ? 1 : (n > 4096.0 ? 4096 : (int)(n + 0.5));
}

// deterministic mode: the launch's slabs must fit the caller's workspace; note how many it used
static int part_claim(FsPartHost* ph, int nslab, FsPart& out) {
  out.base = nullptr; out.stride = 0;
  if (ph == nullptr || ph->base == nullptr) return FS_OK;
  if (nslab > ph->cap) return FS_ERR_ARG;
  if (nslab > ph->used) ph->used = nslab;
  out.base = ph->base; out.stride = ph->stride;
  return FS_OK;
}

template <class P, int WM, int WN, int MI, int NI>
int launch_linear_wgrad(LwArgs a, int target, FsPartHost* ph, hipStream_t stream) {
  a.tiles_ci = cdiv(a.Cin, WM * MI * 32); a.tiles_co = cdiv(a.Cout, WN * NI * 32);
  const int ntile = a.tiles_ci * a.tiles_co;
  int nsplit = target / ntile;
  if (nsplit < 1) nsplit = 1;
  {
    const int nm = splitk_model(cdiv(a.rows, 32), ntile, 1.0, WM * MI * WN * NI * 4.0);      // a 32-row chunk ~ 1 us; tile = 32 x 32 x 4 B per block pair
    if (nm < nsplit) nsplit = nm;
  }
  a.rows_per_split = cdiv(cdiv(a.rows, nsplit), 32) * 32;
  nsplit = cdiv(a.rows, a.rows_per_split);
  if (part_claim(ph, nsplit, a.part) != FS_OK) return FS_ERR_ARG;
  if (a.part.base == nullptr) a.bpart = nullptr;
  const int ntap = a.ntap > 0 ? a.ntap : 1;
  hipLaunchKernelGGL((linear_wgrad_kernel<P, WM, WN, MI, NI>), dim3((unsigned)(ntile * nsplit * ntap)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// Tile and split choice, from a sweep over the SegFormer-B5 and HRNet 1x1 layers (profiles/r03/linear_wgrad_sweep.txt): narrow operands
// take a 64-wide tile on their side; wide layers take 128 x 128 over 512 workgroups while a split still sees >= 512 rows (the row loop
// dominates), otherwise 64 x 64 over 1024 (a quarter of the atomics per workgroup: the epilogue dominates).
int run_linear_wgrad(LwArgs l, FsPartHost* ph, hipStream_t stream) {
  static const int force = FS_ENV_INT("FS_LW_TILE", 0);       // kernel A/B builds only (common.h), read once
  static const int force_wgs = FS_ENV_INT("FS_LW_WGS", 0);
  int tile, target = 512;
  if (l.Cin <= 64 && l.Cout <= 64) tile = 1;
  else if (l.Cin <= 64) tile = 2;
  else if (l.Cout <= 64) tile = 3;
  else {
    const int ntile = cdiv(l.Cin, 128) * cdiv(l.Cout, 128);
    const int nsplit = 512 / ntile > 0 ? 512 / ntile : 1;
    tile = l.rows / nsplit >= 512 ? 4 : 1;
    if (tile == 1) target = 1024;
  }
  if (l.ntap > 0) target = 1024 / l.ntap > 64 ? 1024 / l.ntap : 64;      // gathered rows: ~1000 workgroups over all taps together
  if (force >= 1 && force <= 4) tile = force;
  if (force_wgs > 0) target = force_wgs;
  switch (tile) {
    case 1: return launch_linear_wgrad<PrecX3, 2, 2, 1, 1>(l, target, ph, stream);      //  64 x  64
    case 2: return launch_linear_wgrad<PrecX3, 1, 4, 2, 1>(l, target, ph, stream);      //  64 x 128
    case 3: return launch_linear_wgrad<PrecX3, 4, 1, 1, 2>(l, target, ph, stream);      // 128 x  64
    default: return launch_linear_wgrad<PrecX3, 2, 2, 2, 2>(l, target, ph, stream);     // 128 x 128
  }
}

// Ph rows x PP pairs <= 32 pairs, (Ph + 2) PP <= 48 slots, PP >= 1: fewest patches, then largest fill
void choose_wgrad_wino_patch(int H, int W, int& Ph, int& PP) {
  const int wp = W / 2;
  long best = -1;
  Ph = 1; PP = 1;
  for (int pp = 1; pp <= 32 && pp <= wp; ++pp)
    for (int ph = 1; ph <= 32 && ph <= H + 1; ++ph) {
      if (ph * pp > WP_PAIRS || (ph + 2) * pp > WP_SLOTS) continue;
      const long patches = (long)cdiv(H, ph) * cdiv(wp, pp);
      const long cost = patches * (cdiv(ph * pp, 16) * 2 + 1) * 1000 - ph * pp;      // k-steps of 16 pairs + a fixed cost per patch
      if (best < 0 || cost < best) { best = cost; Ph = ph; PP = pp; }
    }
}

// Patch choice: Ph*Pw <= 64 pixels (padded to a multiple of 16 for the k-steps), halo (Ph+2)(Pw+2) <= 112 slots;
// minimise patches * (k-steps per patch + 1.5).
void choose_wgrad_patch(int H, int W, int NR, int NS, int& Ph, int& Pw) {
  long best = -1;
  Ph = 8; Pw = 8;
  for (int pw = 2; pw <= 64 && pw <= W + 1; ++pw)
    for (int ph = 1; ph <= 64 && ph <= H + 1; ++ph) {
      if (ph * pw > YP || (ph + NR - 1) * (pw + NS - 1) > XS) continue;
      const long patches = (long)cdiv(H, ph) * cdiv(W, pw);
      // measured on 20x20 x 256 ch: time ~ patches * (k-steps per patch + 1.5) -- every patch pays a load/split/barrier round
      const long cost = patches * (2 * cdiv(ph * pw, 16) + 3) * 10000 + ((pw & 3) ? 5000 : 0) + (ph + NR - 1) * (pw + NS - 1);
      if (best < 0 || cost < best) { best = cost; Ph = ph; Pw = pw; }
    }
}

}  // namespace

namespace {
template <class P, int NR, int NS>
int launch_class(WgArgs a, int ntile, FsPartHost* ph, hipStream_t stream) {
  choose_wgrad_patch(a.H, a.W, NR, NS, a.Ph, a.Pw);
  a.tiles_y = cdiv(a.H, a.Ph); a.tiles_x = cdiv(a.W, a.Pw);
  a.magic_wh = div_magic1(a.Pw + NS - 1); a.magic_pw = div_magic1(a.Pw);
  a.npatch = a.B * a.tiles_y * a.tiles_x;
  // 512 patch streams = 8 waves on every CU, never a short second round: 512 workgroups (two per CU).
  // (A wave-specialised variant -- 4 producer + 4 consumer waves, double-buffered LDS, one workgroup per CU -- measured
  // +14 % on this kernel alone and -1 % on the training step, where kernels of other HRNet branches share the CUs.)
  int nsplit = 512 / ntile;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.npatch) nsplit = a.npatch;
  {
    const int nm = splitk_model(a.npatch, ntile, 0.75 * NR * NS, 16.0 * NR * NS);      // ~0.75 us and 16 KB per tap and patch round
    if (nm < nsplit) nsplit = nm;
  }
  a.patches_per_split = cdiv(a.npatch, nsplit);
  nsplit = cdiv(a.npatch, a.patches_per_split);
  if (part_claim(ph, nsplit, a.part) != FS_OK) return FS_ERR_ARG;
  {
    // kernel A/B builds only.  Measured (profiles/r04/wgrad_phase_trace.txt): the rounds of the two workgroups of a CU even out (14 500 /
    // 25 000 -> 15 900 cycles each) and the MFMA loops of the launch end 18 % earlier -- but then all 512 workgroups reach their split-K
    // atomics together instead of half of them early, the launch takes 4 % LONGER alone and the training step is unchanged (the
    // step follows its switching work, not its stalls: DESIGN.md 4c): off.
    static const int prio = FS_ENV_INT("FS_WGRAD_PRIO", 0);
    a.prio = prio;
  }
#ifdef FS_WGRAD_TRACE
  static long long* dbg = nullptr;
  const long nwg = (long)ntile * nsplit;
  const bool traced = NR == 3 && NS == 3 && nwg <= 4096 && a.patches_per_split >= 6;
  if (traced) {
    if (dbg == nullptr && hipMalloc(&dbg, sizeof(long long) * 4096 * 4 * 4 * 8) != hipSuccess) return FS_ERR_ARG;
    if (hipMemsetAsync(dbg, 0, sizeof(long long) * nwg * 4 * 4 * 8, stream) != hipSuccess) return FS_ERR_ARG;
  }
  a.dbg = traced ? dbg : nullptr;
#endif
  hipLaunchKernelGGL((conv_wgrad_class_kernel<P, NR, NS>), dim3((unsigned)(ntile * nsplit)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
#ifdef FS_WGRAD_TRACE
  if (traced) {
    static long long host[4096 * 4 * 4 * 8];
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(host, dbg, sizeof(long long) * nwg * 128, hipMemcpyDeviceToHost) != hipSuccess) return FS_ERR_ARG;
    // phases: [0->1] address + issue loads, [1->2] barrier (partners still multiplying), [2->3] loads landed, [3->4] split + LDS stores,
    // [4->5] barrier, [5->6] MFMA loop, [6->0'] loop overhead to the next round's top; per wave, rounds 2..4 (round 5 has no successor stamp)
    double sum[8] = {0}, skew_mfma = 0, round_len = 0; long n = 0, nr = 0;
    double par_mfma[2] = {0, 0}, par_round[2] = {0, 0}, par_start[2] = {0, 0}; long par_n[2] = {0, 0};
    long long tmin = 0;
    for (long w_ = 0; w_ < nwg; ++w_) { const long long v = host[(w_ * 4 * 4) * 8]; if (v != 0 && (tmin == 0 || v < tmin)) tmin = v; }
    for (long w_ = 0; w_ < nwg; ++w_) {
      for (int it = 0; it < 3; ++it) {
        long long end_min = 0, end_max = 0;
        for (int wv = 0; wv < 4; ++wv) {
          const long long* t = host + ((w_ * 4 + wv) * 4 + it) * 8;
          const long long* tn = t + 8;
          if (t[0] == 0 || t[6] == 0 || tn[0] == 0) continue;
          for (int i = 0; i < 6; ++i) sum[i] += (double)(t[i + 1] - t[i]);
          sum[6] += (double)(tn[0] - t[6]);
          round_len += (double)(tn[0] - t[0]);
          {
            const int par = (int)((unsigned)host[((w_ * 4 + wv) * 4) * 8 + 7] & 1u);
            par_mfma[par] += (double)(t[6] - t[5]); par_round[par] += (double)(tn[0] - t[0]); ++par_n[par];
            if (it == 0) par_start[par] += (double)(t[0] - tmin);
          }
          ++n;
          if (wv == 0 || t[6] < end_min) end_min = t[6];
          if (wv == 0 || t[6] > end_max) end_max = t[6];
        }
        skew_mfma += (double)(end_max - end_min); ++nr;
      }
    }
    {
      long hist[16] = {0}; long same = 0, pairs = 0;
      static int cu_slot[8][16][16][4];      // [xcc guess = wg % 8][se][cu][simd] -> last wave_id seen
      for (auto& a0 : cu_slot) for (auto& a1 : a0) for (auto& a2 : a1) for (int& v : a2) v = -1;
      for (long w_ = 0; w_ < nwg; ++w_)
        for (int wv = 0; wv < 4; ++wv) {
          const unsigned id = (unsigned)host[((w_ * 4 + wv) * 4) * 8 + 7];
          const int wid = id & 15, simd = (id >> 4) & 3, cu = (id >> 8) & 15, se = (id >> 13) & 7;
          ++hist[wid];
          int& prev = cu_slot[w_ % 8][se][cu][simd];
          if (prev >= 0) { ++pairs; if ((prev & 1) == (wid & 1)) ++same; }
          prev = wid;
        }
      fprintf(stderr, "wgrad trace: wave_id histogram");
      for (int i = 0; i < 16; ++i) if (hist[i]) fprintf(stderr, " %d:%ld", i, hist[i]);
      fprintf(stderr, " | SIMDs with two traced waves %ld, of them with EQUAL slot parity %ld\n", pairs, same);
    }
    if (n > 0) {
      fprintf(stderr, "wgrad trace B%d %dx%d %d->%d patches/split %d grid %ld: round %.0f cyc =", a.B, a.H, a.W, a.Cin, a.Cout, a.patches_per_split, nwg, round_len / n);
      const char* nm[7] = {"issue", "barrier1", "loads", "split", "barrier2", "mfma", "next"};
      for (int i = 0; i < 7; ++i) fprintf(stderr, " %s %.0f", nm[i], sum[i] / n);
      fprintf(stderr, " | spread of the four waves' MFMA-loop ends %.0f", skew_mfma / nr);
      for (int par = 0; par < 2; ++par)
        if (par_n[par]) fprintf(stderr, " | slot %d: mfma %.0f round %.0f round-2 top at +%.0f", par, par_mfma[par] / par_n[par], par_round[par] / par_n[par], par_start[par] * 3 / par_n[par]);
      fprintf(stderr, "\n");
    }
  }
#endif
  return FS_OK;
}

// Plane table of a 3x3 filter with stride st (2 or 3) for a Ph x Pw patch; returns the slot count
static int planes_plan(MpArgs& a) {
  const int st = a.st, nres = st < 3 ? st : 3;
  int n = 0, base = 0;
  int idx[3][3];
  for (int pr = 0; pr < nres; ++pr)
    for (int pc = 0; pc < nres; ++pc) {
      const int extra_r = (2 - pr) / st, extra_c = (2 - pc) / st;      // largest (r div st) over taps r = pr, pr + st, ... <= 2
      a.pl_base[n] = base; a.pl_wcol[n] = a.Pw + extra_c; a.pl_pr[n] = pr; a.pl_pc[n] = pc; a.pl_magic[n] = div_magic1(a.Pw + extra_c);
      base += (a.Ph + extra_r) * (a.Pw + extra_c);
      idx[pr][pc] = n++;
    }
  for (int k = n; k < 9; ++k) { a.pl_base[k] = 1 << 30; a.pl_wcol[k] = 1; a.pl_pr[k] = 0; a.pl_pc[k] = 0; a.pl_magic[k] = 0u; }
  a.nplanes = n; a.nslots = base;
  for (int r = 0; r < 3; ++r)
    for (int s2 = 0; s2 < 3; ++s2) {
      const int p = idx[r % st][s2 % st];
      a.tap_off[r * 3 + s2] = (a.pl_base[p] + (r / st) * a.pl_wcol[p] + s2 / st) * 64;
      a.tap_w[r * 3 + s2] = a.pl_wcol[p] == a.Pw + 1 ? 1 : 0;
    }
  return base;
}

// Ph x Pw <= 32 dY pixels, all planes together <= 160 slots: fewest patches x (k-steps + 1.5), as choose_wgrad_patch
static bool choose_planes_patch(MpArgs& a) {
  long best = -1;
  int bh = 0, bw = 0;
  for (int pw = 2; pw <= MP_YP && pw <= a.W + 1; ++pw)
    for (int ph = 1; ph <= MP_YP && ph <= a.H + 1; ++ph) {
      if (ph * pw > MP_YP) continue;
      a.Ph = ph; a.Pw = pw;
      if (planes_plan(a) > MP_XS) continue;
      const long patches = (long)cdiv(a.H, ph) * cdiv(a.W, pw);
      const long cost = patches * (2 * cdiv(ph * pw, 16) + 3) * 10000 + ((pw & 3) ? 5000 : 0) + a.nslots;
      if (best < 0 || cost < best) { best = cost; bh = ph; bw = pw; }
    }
  if (best < 0) return false;
  a.Ph = bh; a.Pw = bw;
  planes_plan(a);
  return true;
}

template <class P>
int launch_planes(MpArgs a, FsPartHost* ph, hipStream_t stream) {
  if (!choose_planes_patch(a)) return FS_ERR_ARG;
  a.tiles_y = cdiv(a.H, a.Ph); a.tiles_x = cdiv(a.W, a.Pw);
  a.magic_pw = div_magic1(a.Pw);
  a.npatch = a.B * a.tiles_y * a.tiles_x;
  const int ntile = a.tiles_ci * a.tiles_co;
  static const int target = FS_ENV_INT("FS_WGRAD_S2_WGS", 512);      // kernel A/B builds only
  int nsplit = target / ntile;
  if (nsplit < 1) nsplit = 1;
  if (nsplit > a.npatch) nsplit = a.npatch;
  a.patches_per_split = cdiv(a.npatch, nsplit);
  nsplit = cdiv(a.npatch, a.patches_per_split);
  if (part_claim(ph, nsplit, a.part) != FS_OK) return FS_ERR_ARG;
  hipLaunchKernelGGL((conv_wgrad_planes_kernel<P>), dim3((unsigned)(ntile * nsplit)), dim3(256), 0, stream, a);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

template <class P>
int run_classes(WgArgs& a, int ntile, int R, int S, int stride, int pad, FsPartHost* ph, hipStream_t stream) {
  for (int r0 = 0; r0 < stride && r0 < R; ++r0)
    for (int s0 = 0; s0 < stride && s0 < S; ++s0) {
      const int nR = (R - r0 + stride - 1) / stride, nS = (S - s0 + stride - 1) / stride;
      a.cy = r0 - pad; a.cx = s0 - pad; a.rbase = r0; a.rstep = stride; a.sbase = s0; a.sstep = stride;
      int e = FS_ERR_ARG;
      if (nR == 3 && nS == 3) e = launch_class<P, 3, 3>(a, ntile, ph, stream);
      else if (nR == 2 && nS == 2) e = launch_class<P, 2, 2>(a, ntile, ph, stream);
      else if (nR == 2 && nS == 1) e = launch_class<P, 2, 1>(a, ntile, ph, stream);
      else if (nR == 1 && nS == 2) e = launch_class<P, 1, 2>(a, ntile, ph, stream);
      else if (nR == 1 && nS == 1) e = launch_class<P, 1, 1>(a, ntile, ph, stream);
      if (e != FS_OK) return e;
    }
  return FS_OK;
}
}  // namespace

// dW and dbias of a linear layer in one launch (bf16x3 only; include/fovealseg.h fs_linear_bwd_weight_bias)
bool fs_linear_wgrad_eligible(int mode, long rows, int Cin, int Cout) {
  return mode == 1 && rows > 0 && Cin % 4 == 0 && Cout % 4 == 0 && Cin >= 16 && Cout >= 16 &&
         (size_t)rows * Cin * 4 < 4294967000UL && (size_t)rows * Cout * 4 < 4294967000UL;
}
int fs_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, long rows, int Cin, int Cout, FsPartHost* part, float* bpart,
                    hipStream_t stream) {
  LwArgs l;
  l.x = x; l.dy = dy; l.dw = dw; l.dbias = dbias; l.rows = (int)rows; l.Cin = Cin; l.Cout = Cout;
  l.x_bytes = (unsigned)((size_t)rows * Cin * 4); l.dy_bytes = (unsigned)((size_t)rows * Cout * 4);
  l.part = FsPart{nullptr, 0}; l.bpart = dbias != nullptr ? bpart : nullptr;
  l.ntap = 0; l.S = 1; l.Ho = l.Wo = l.Hx = l.Wx = 1; l.st = 1; l.pad = 0;
  return run_linear_wgrad(l, part, stream);
}

// strided 3x3 layers whose bwd-weight runs as nine gathered-row GEMMs (policy above); conv.hip keeps them off the store + reduce route
bool fs_wgrad_gather_s2(int Cin, int R, int S, int stride) { return R == 3 && S == 3 && (stride == 2 || stride == 3) && Cin <= 64; }

bool fs_wgrad_split_eligible(int Cin, int Cout, int R, int S, int stride, int pad, int dil) {
  (void)pad;
  if (dil != 1 || Cin % 4 || Cout % 4 || Cin < 16 || Cout < 16 || R != S) return false;
  if (R == 3 && stride == 1) return true;                                   // one class of 3 x 3 taps
  const int nr = (R + stride - 1) / stride;                                // taps per class and dimension
  return nr <= 2 && (stride < R ? stride : R) <= 3;                        // classes of 1 or 2 taps per dimension, at most 9 classes
}

// dW of any conv2d with square filter: one launch per tap class (dw zeroed by the caller or accumulated into).  mode: 1 = bf16x3, 2 = f16x2
int fs_wgrad_split(int mode, const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S,
                   int stride, int pad, FsPartHost* ph, hipStream_t stream) {
  if ((size_t)B * H * W * Cin * 4 >= 4294967000UL || (size_t)B * Ho * Wo * Cout * 4 >= 4294967000UL) return FS_ERR_ARG;
  WgArgs a;
  a.part = FsPart{nullptr, 0};
  a.x = x; a.dy = dy; a.dw = dw;
  a.B = B; a.H = Ho; a.W = Wo; a.Hx = H; a.Wx = W; a.Cin = Cin; a.Cout = Cout;
  a.sm = stride; a.S = S;
  a.tiles_ci = cdiv(Cin, 64); a.tiles_co = cdiv(Cout, 64);
  a.x_bytes = (unsigned)((size_t)B * H * W * Cin * 4);
  a.dy_bytes = (unsigned)((size_t)B * Ho * Wo * Cout * 4);
  const int ntile = a.tiles_ci * a.tiles_co;
  static const int wino_pol = FS_ENV_INT("FS_WGRAD_WINO", 1);      // kernel A/B builds only: 0 never, 2 always
  const bool wino_wgrad = wino_pol != 0, wino_wgrad_all = wino_pol == 2;
  if (mode == 1 && wino_wgrad && R == 3 && S == 3 && stride == 1 && pad == 1 && H == Ho && W == Wo && W % 2 == 0 && W >= 2) {
    // bf16x3, even width: the transform-domain kernel (2/3 of the MFMAs)
    WwArgs w;
    w.x = x; w.dy = dy; w.dw = dw; w.B = B; w.H = H; w.W = W; w.Cin = Cin; w.Cout = Cout;
    choose_wgrad_wino_patch(H, W, w.Ph, w.PP);
    w.tiles_y = cdiv(H, w.Ph); w.tiles_x = cdiv(W / 2, w.PP);
    w.npatch = B * w.tiles_y * w.tiles_x;
    w.tiles_ci = a.tiles_ci; w.tiles_co = a.tiles_co;
    w.x_bytes = a.x_bytes; w.dy_bytes = a.dy_bytes;
    w.magic_pp = div_magic1(w.PP);
    int nsplit = 256 / ntile;                  // one 512-thread workgroup per CU
    if (nsplit < 1) nsplit = 1;
    if (nsplit > w.npatch) nsplit = w.npatch;
    w.patches_per_split = cdiv(w.npatch, nsplit);
    nsplit = cdiv(w.npatch, w.patches_per_split);
    // Measured against conv_wgrad_class_kernel in one gpurun call (us, bf16x3, B = 64): 64->64 @ 80x80 207 vs 183, 128->128 @ 40x40 195 vs
    // 178, 256->256 @ 20x20 218 vs 191, 512->512 @ 10x10 217 vs 209 -- SLOWER where a workgroup sees few patches (one 8-wave workgroup per
    // CU pays more per barrier round than two independent 4-wave ones, and the doubled split work is not hidden) -- but 960->240 @ 80x80
    // 7.0 vs 8.1 ms: long pixel loops per channel tile.  So only layers with >= 64 patches per workgroup come here (the C1 heads: 960 -> 240 at B >= 4).
    if (w.patches_per_split < 64 && !wino_wgrad_all) goto direct;
    if (part_claim(ph, 2 * nsplit, w.part) != FS_OK) return FS_ERR_ARG;
    constexpr int lds = PrecX3::NPL * (XT_PLANE + DM_PLANE);
    {
      static unsigned long long done = 0ull;
      int dev = 0;
      if (hipGetDevice(&dev) != hipSuccess) return FS_ERR_ARG;
      if (dev < 0 || dev >= 64 || !((done >> dev) & 1ull)) {
        const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_wgrad_wino_kernel<PrecX3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (attr != hipSuccess) return (int)attr;
        if (dev >= 0 && dev < 64) done |= 1ull << dev;
      }
    }
    hipLaunchKernelGGL((conv_wgrad_wino_kernel<PrecX3>), dim3((unsigned)(ntile * nsplit)), dim3(512), lds, stream, w);
    FS_LAUNCH_CHECK();
    return FS_OK;
  }
direct:
  static const bool linear_on = FS_ENV_INT("FS_WGRAD_LINEAR", 1) != 0;      // kernel A/B builds only
  if (mode == 1 && linear_on && R == 1 && S == 1 && stride == 1 && pad == 0 && H == Ho && W == Wo) {
    LwArgs l;
    l.x = x; l.dy = dy; l.dw = dw; l.dbias = nullptr; l.rows = B * H * W; l.Cin = Cin; l.Cout = Cout;
    l.x_bytes = a.x_bytes; l.dy_bytes = a.dy_bytes;
    l.part = FsPart{nullptr, 0}; l.bpart = nullptr;
    l.ntap = 0; l.S = 1; l.Ho = l.Wo = l.Hx = l.Wx = 1; l.st = 1; l.pad = 0;
    return run_linear_wgrad(l, ph, stream);
  }
  // ... and so do strided 3x3 layers with <= 64 input channels: there the nine-accumulator kernels are bound by their split-K reduction
  // (512 x 147 KB), while nine GEMMs over gathered rows have 32 KB tiles, prefetch their next chunk during the MFMA phase and read X only
  // 2.25 times (each tap a quarter of the pixels).  Same-box A/B (profiles/r04/wgrad_gather_s2_ab.txt, us): 64 -> 64 @ 80x80 104 -> 75,
  // 64 -> 128 173 -> 138, 64 -> 256 @ 40x40 100 -> 75; 128 -> 256 155 -> 151, 256 -> 512 157 -> 162, 512 -> 512 268 -> 356 (those stay).
  static const int gather_pol = FS_ENV_INT("FS_WGRAD_GATHER", 1);      // kernel A/B builds only: 0 off, 2 = every strided layer
  const bool gather_s2 = stride > 1 && (gather_pol == 2 || fs_wgrad_gather_s2(Cin, R, S, stride)) && !(ph != nullptr && ph->force_planes);
  if (mode == 1 && gather_pol != 0 && ((stride >= R && stride >= S) || gather_s2) && R * S <= 16 &&
      (R > 1 || stride > 1) && (long)B * Ho * Wo < 2000000000L / (Cin > Cout ? Cin : Cout)) {
    // every tap class is a single tap (3x3 stride 4, 1x1 stride 2 ...): R * S linear layers over gathered X rows, one launch
    LwArgs l;
    l.x = x; l.dy = dy; l.dw = dw; l.dbias = nullptr; l.rows = B * Ho * Wo; l.Cin = Cin; l.Cout = Cout;
    l.x_bytes = a.x_bytes; l.dy_bytes = a.dy_bytes;
    l.part = FsPart{nullptr, 0}; l.bpart = nullptr;
    l.ntap = R * S; l.S = S; l.Ho = Ho; l.Wo = Wo; l.Hx = H; l.Wx = W; l.st = stride; l.pad = pad;
    return run_linear_wgrad(l, ph, stream);
  }
  // Measured in one gpurun call against the per-class launches (profiles/r04/wgrad_s2_planes_ab.txt, us, bf16x3, B = 64): 64 -> 64 @ 80x80
  // 164 -> 123, but 64 -> 128 @ 80x80 158-165 -> 166-172, 128 -> 256 @ 40x40 167 -> 174, 256 -> 512 @ 20x20 176 -> 177: the launch is one
  // round of 512 workgroups whose time is (patch rounds per workgroup) x ~7 us of load -> split -> barrier -> MFMA latency plus ~80 us
  // of split-K atomics (512 x 147 KB at the L2's one dword per channel and clock), and neither term depends on how the taps are grouped;
  // only a single channel tile (twice the splits, half the rounds) gains.  planes policy 1 = those layers, 2 = every strided 3x3, 0 = off.
  static const int planes_pol = FS_ENV_INT("FS_WGRAD_PLANES", 1);      // kernel A/B builds only
  if (((planes_pol != 0 && (planes_pol == 2 || ntile == 1)) || (ph != nullptr && ph->force_planes)) && R == 3 && S == 3 && (stride == 2 || stride == 3)) {
    // all nine taps of a strided 3x3 filter in one launch (parity planes of the X halo in LDS)
    MpArgs m;
    m.x = x; m.dy = dy; m.dw = dw; m.B = B; m.H = Ho; m.W = Wo; m.Hx = H; m.Wx = W; m.Cin = Cin; m.Cout = Cout;
    m.st = stride; m.pad = pad; m.tiles_ci = a.tiles_ci; m.tiles_co = a.tiles_co;
    m.x_bytes = a.x_bytes; m.dy_bytes = a.dy_bytes; m.part = FsPart{nullptr, 0};
    return mode == 2 ? launch_planes<PrecF16>(m, ph, stream) : launch_planes<PrecX3>(m, ph, stream);
  }
  return mode == 2 ? run_classes<PrecF16>(a, ntile, R, S, stride, pad, ph, stream) : run_classes<PrecX3>(a, ntile, R, S, stride, pad, ph, stream);
}
