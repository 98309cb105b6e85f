// Internal interface between conv.hip (C-ABI entry points, kernel selection) and the split-precision kernel families
// (conv_halo.hip, conv_tapset.hip, conv_wgrad.hip).  Not part of include/fovealseg.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// ---- conv_halo.hip: halo-tiled 3x3 / stride 1 / pad 1 kernel.  mode: 1 = bf16x3, 2 = f16x2 (conv_split.h) ----------------
// 3x3, stride 1, pad 1, dilation 1, channel counts multiples of 4, K >= 32
bool fs_halo_eligible(int H, int W, int Cs, int Cd, int R, int S, int stride, int pad, int dil);
// bytes of the pre-split weight pack (header included) for K = Cs source channels and N = Cd destination channels
long fs_halo_pack_bytes(int mode, int Cs, int Cd);
// number of pixel tiles (= BatchNorm partial-sum slabs) the kernel uses for a (B,H,W) output
int fs_halo_stats_slabs(int B, int H, int W);
// pack w (RSCK fp32, logical Cin x Cout) into ws, then run the conv.  transposed = 1: bwd-data (src = dY with Cs = Cout channels,
// dst = dX with Cd = Cin channels).  stats may be null.  w_amax: max|w| bits kept by the caller, or null (f16x2 only).
int fs_halo_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                    int B, int H, int W, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh,
                    uint32_t drop_key, hipStream_t stream);
// where the f16x2 kernels read max |w| (float bits) of a weight tensor: w_amax when the caller maintains it, else the first
// word of ws, filled here by a memset + atomic-max kernel.
const unsigned* fs_f16_weight_amax(const float* w, long n, void* ws, const unsigned* w_amax, hipStream_t stream, int* err);
// How the calling thread's next conv entry point treats `ws` (fs_conv2d_ws_mode / fs_conv2d_pack, include/fovealseg.h): 0 = pack the
// weights into ws, then run (the default); FS_WS_RUN_ONLY = ws already holds this layer's pack for this shape and precision (the pack
// launch is skipped); FS_WS_PACK_ONLY = run the pack launch and return.  Every kernel family with a weight pack honours it.
extern thread_local int fs_ws_mode_tls;
#define FS_WS_RUN_ONLY 1
#define FS_WS_PACK_ONLY 2
int fs_weight_amax_segments_impl(const float* arena, const long* offsets, const long* sizes, int nparams, unsigned* out, hipStream_t stream);

// ---- conv_wino.hip: the same problem class with F(2,3) minimal filtering along the row (12 MFMA steps per pixel pair instead of 18).
// Even W >= 4, B*(H+1) < 65536, channel counts multiples of 4, K >= 32 (K >= 128 in f16x2); FS_WINOGRAD=0 switches it off.
// optional epilogue mode of the bwd-data call: BatchNorm-backward column sums of the layer whose output gradient is being written
struct FsBnSums {
  const float* y; const unsigned char* mask; const float* mean; const float* invstd;      // bwd-data: BatchNorm-backward sums of the consumer layer
  const float* add_src; const unsigned char* add_mask;                                    // bwd-data: second gradient joining in the epilogue
  const float* ep_scale; const float* ep_shift; const float* ep_res; int ep_act;          // forward (inference): affine + residual + activation
  // forward of a linear layer that ends a residual branch (1x1 GEMM kernel, round 5): dst = add_src + DropPath(Dropout(conv + bias)) --
  // rows [b * dp_rows, (b + 1) * dp_rows) belong to sample b, kept with the sample hash of fs_residual_droppath (dp_thresh 0: no DropPath)
  float dp_scale; uint32_t dp_thresh, dp_key; int dp_rows;
};
bool fs_wino_eligible(int mode, int B, int H, int W, int Cs, int Cd);
long fs_wino_pack_bytes(int mode, int Cs, int Cd);
int fs_wino_stats_slabs(int mode, int B, int H, int W, int Cs, int Cd);
int fs_wino_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                    int B, int H, int W, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh,
                    uint32_t drop_key, const FsBnSums* bn, hipStream_t stream);

// ---- conv_wino4.hip (round 5): the same problem class with F(4,3) along the row (18 MFMA steps per four output pixels instead of 24):
// bf16x3 only, W a multiple of 4.  fs_wino_conv3x3 / fs_wino_pack_bytes / fs_wino_stats_slabs route to it where fs_wino4_selected holds.
bool fs_wino_takes_f43(int mode, int B, int H, int W, int Cs, int Cd);      // conv_wino.hip: this problem goes to the F(4,3) kernel
bool fs_wino4_eligible(int mode, int B, int H, int W, int Cs, int Cd);
long fs_wino4_pack_bytes(int mode, int Cs, int Cd);
int fs_wino4_stats_slabs(int B, int H, int W, int Cs, int Cd);
int fs_wino4_conv3x3(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, int B, int H, int W, int Cs,
                     int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh, uint32_t drop_key, const FsBnSums* bn,
                     hipStream_t stream);

// ---- conv_pointwise.hip: 1x1 / stride 1 / pad 0 as a GEMM with pre-split weights (forward and bwd-data).  mode: 1 = bf16x3, 2 = f16x2 ----
bool fs_pointwise_eligible(int Cs, int Cd, int R, int S, int stride, int pad, int dil);
long fs_pointwise_pack_bytes(int mode, int Cs, int Cd);
// forward of a stride >= filter convolution as one GEMM over gathered rows (K = R*S*Cin; pack bytes = fs_pointwise_pack_bytes(mode, K, Cout))
bool fs_pointwise_gather_eligible(int Cin, int Cout, int R, int S, int stride, int dil);
int fs_pointwise_gather_conv(int mode, const float* x, const float* w, const float* bias, float* y, float* stats, void* ws, const unsigned* w_amax,
                             int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, float drop_scale,
                             uint32_t drop_thresh, uint32_t drop_key, hipStream_t stream);
// bwd-data of those layers: one GEMM per tap whose rows are scattered to the tap's residue class of dX (the caller zero-fills the classes no
// tap reaches); pack bytes = fs_pointwise_pack_bytes(mode, Cout, Cin)
bool fs_pointwise_scatter_eligible(int Cin, int Cout, int R, int S, int stride, int dil);
int fs_pointwise_scatter_conv(int mode, const float* dy, const float* w, float* dx, void* ws, const unsigned* w_amax, int B, int H, int W,
                              int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, hipStream_t stream);
// M = B*H*W rows; transposed = 1: bwd-data (src = dY with Cs = Cout channels, dst = dX with Cd = Cin).  stats: [ceil(M/128)][Cd][2] or null.
// bn (bwd-data only, may be null): stats then receives the BatchNorm-backward sums of the layer that produced x, and / or a second gradient joins dst
int fs_pointwise_conv(int mode, const float* src, const float* w, const float* bias, float* dst, float* stats, void* ws, const unsigned* w_amax,
                      long M, int Cs, int Cd, int Cin, int Cout, int transposed, float drop_scale, uint32_t drop_thresh, uint32_t drop_key,
                      const FsBnSums* bn, hipStream_t stream);
int fs_pointwise_stats_slabs(long M);

// ---- deterministic split-K (include/fovealseg.h fs_set_deterministic) ----------------------------------------------------------
// Every bwd-weight kernel ends by adding its workgroup's partial dW tile to the tiles of the other pixel splits.  By default that is an
// fp32 atomic per element (order = arrival order: results differ in the last bits from run to run).  In deterministic mode each split
// writes its partial tile with plain stores into ITS OWN slab (a full dW image) of the caller's workspace and fs_wgrad_reduce sums the
// slabs in index order.  base == nullptr selects the atomics.
struct FsPart { float* base; long stride; };
__device__ __forceinline__ void fs_wgrad_out(float* dw, const FsPart p, int slab, long idx, float v) {
#ifdef FS_WGRAD_NT
  // kernel A/B builds only (-DFS_WGRAD_NT): the partial tiles leave with non-temporal stores, so that 75 MB of slab lines per launch do
  // not displace the layer's dY / X from L2 between its bwd-weight and its bwd-data (measured in profiles/r05/wgrad_nt_ab.txt)
  if (p.base != nullptr) __builtin_nontemporal_store(v, p.base + (long)slab * p.stride + idx);
#else
  if (p.base != nullptr) p.base[(long)slab * p.stride + idx] = v;
#endif
  else atomicAdd(dw + idx, v);
}
bool fs_deterministic();
// dw[i] = (accumulate ? dw[i] : 0) + part[0][i] + part[1][i] + ... + part[nslab-1][i], in that order
int fs_wgrad_reduce(float* part, int nslab, long n, float* dw, int accumulate, hipStream_t stream);
// host-side bookkeeping of one bwd-weight call in deterministic mode: slabs available / slabs the launches used
struct FsPartHost { float* base; long stride; long cap; int used; int force_planes; };      // force_planes: the strided 3x3 layer must take the one-launch kernel (every slab element written: no memset)

// ---- conv_wgrad.hip: split-precision weight gradient, one launch per tap class (dw zeroed by the caller or accumulated into) ----
// any square filter / stride whose tap classes have at most 2 taps per dimension (3x3 s2/s4, 1x1 any stride), plus 3x3 s1
bool fs_wgrad_split_eligible(int Cin, int Cout, int R, int S, int stride, int pad, int dil);
bool fs_wgrad_gather_s2(int Cin, int R, int S, int stride);
bool fs_linear_wgrad_eligible(int mode, long rows, int Cin, int Cout);
// part (nullable): deterministic mode.  fs_linear_wgrad: bpart (nullable) = slabs of Cout floats for the bias column sums, 4 per split.
int fs_linear_wgrad(const float* x, const float* dy, float* dw, float* dbias, long rows, int Cin, int Cout, FsPartHost* part, float* bpart,
                    hipStream_t stream);
int fs_wgrad_split(int mode, const float* x, const float* dy, float* dw, int B, int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S,
                   int stride, int pad, FsPartHost* part, hipStream_t stream);

// ---- conv_s2bwd.hip: bwd-data of a 3x3 / stride 2 / pad 1 convolution, the four output parities in one launch ----
bool fs_s2bwd_eligible(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil);
long fs_s2bwd_pack_bytes(int mode, int Cin, int Cout);
// bn (may be null): slab[fs_s2bwd_stats_slabs][Cin][2] receives the BatchNorm-backward sums of the layer that produced x, and / or a second gradient joins dX
int fs_s2bwd_stats_slabs(int B, int Ho, int Wo);
int fs_s2bwd_conv(int mode, const float* dy, const float* w, float* dx, void* ws, const unsigned* w_amax, int B, int H, int W, int Cin, int Ho,
                  int Wo, int Cout, const FsBnSums* bn, float* slab, hipStream_t stream);

// ---- conv_s2fwd.hip: forward of a 3x3 / stride 2 / pad 1 convolution, the four input parity planes in one LDS refill per chunk ----
bool fs_s2fwd_eligible(int H, int W, int Cin, int Ho, int Wo, int Cout, int R, int S, int stride, int pad, int dil);
long fs_s2fwd_pack_bytes(int mode, int Cin, int Cout);
int fs_s2fwd_slabs(int B, int Ho, int Wo);          // BatchNorm partial-sum slabs (= pixel tiles) of the stats variant
int fs_s2fwd_conv(int mode, const float* x, const float* w, const float* bias, float* y, float* stats, void* ws, const unsigned* w_amax,
                  int B, int H, int W, int Cin, int Ho, int Wo, int Cout, float drop_scale, uint32_t drop_thresh, uint32_t drop_key,
                  hipStream_t stream);

// conv_tapset.hip: general halo-tiled split-precision convolution over a list of tap classes.
//   source row of (loop row oy, class tap tr) = sm*(oy + tr) + cy,  filter row r = rbase + rstep*tr  (columns alike)
struct FsTapClass { int cy, cx, nR, nS, rbase, rstep, sbase, sstep; };
struct FsTapsetProblem {
  const float* src; const float* w; const float* bias; float* dst; float* stats; void* ws; const unsigned* w_amax;
  int B, Hs, Ws, Cs, Hd, Wd, Cd;     // source / destination tensors (NHWC)
  int Cin, Cout, R, S;               // logical weight shape [R][S][Cin][Cout]
  int transposed;                    // 0: K = Cin, N = Cout;  1 (bwd-data): K = Cout, N = Cin
  int Hq, Wq, os, oy0, ox0, sm;      // loop grid; destination pixel = (oy*os + oy0, ox*os + ox0)
  int ncls; FsTapClass cls[9];
  float drop_scale; uint32_t drop_thresh, drop_key;
};
long fs_tapset_pack_bytes(int mode, int Cs, int Cd, int total_taps);
int fs_tapset_slabs(int B, int Hq, int Wq, int maxR, int maxS);
int fs_tapset_conv(int mode, const FsTapsetProblem& p, hipStream_t stream);      // mode: 1 = bf16x3, 2 = f16x2
void fs_tapset_patch(int Hq, int Wq, int maxR, int maxS, int* Ph, int* Pw);
