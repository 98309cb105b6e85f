// Split-precision arithmetic shared by the halo-tiled convolution kernels (conv_halo.hip, conv_tapset.hip, conv_wgrad.hip).
// An fp32 operand reaches the matrix cores as NPL 16-bit planes; a product is the sum of the plane products that matter:
//   PrecF16 ("f16x2"): x * 2^s = h1 + h2 (two fp16, 22 bits), product = h1 g2 + h2 g1 + h1 g1  (3 MFMAs), operands scaled
//   PrecX3 ("bf16x3"): x = x1 + x2 + x3 (three bf16, 24 bits), product = x1y3 + x2y2 + x3y1 + x1y2 + x2y1 + x1y1 (6 MFMAs)
// Both accumulate in fp32 (v_mfma_f32_32x32x16_{f16,bf16}); DESIGN.md "Precision modes".
#pragma once
#include "common.h"

namespace fs_split {

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));

constexpr unsigned OOB = 0xFFFFFFF0u;      // raw-buffer offset past every tensor: the load returns 0, the store is dropped
constexpr int HDR = 256;                   // bytes of pack header (max|w| bits at offset 0 when the library computes them)
constexpr int EMIN = -100;                 // exponent floor (all-zero tiles)

__device__ __forceinline__ __amdgpu_buffer_rsrc_t make_rsrc(const void* p, unsigned bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p), 0, (int)bytes, 0x00020000);
}
__device__ __forceinline__ float pow2f(int e) {          // 2^e, 0 below the normal range
  return e < -126 ? 0.f : __builtin_bit_cast(float, (unsigned)(e + 127) << 23);
}
__device__ __forceinline__ int exponent_of_bits(unsigned bits) {
  const int e = (int)((bits >> 23) & 0xffu) - 127;
  return e < EMIN ? EMIN : e;
}
// n / d for 0 <= n < 65536 with m = 2^32 / d + 1 precomputed on the host (exact for d <= 65535): two VALU instead of ~25
__device__ __forceinline__ int div_small(int n, unsigned m) { return (int)__umulhi((unsigned)n, m); }
static inline unsigned div_magic(int d) { return (unsigned)(4294967296ULL / (unsigned)d + 1ULL); }
// the same for divisors that may be 1 (magic 0 = identity)
__device__ __forceinline__ int div_small1(int n, unsigned m) { return m == 0u ? n : (int)__umulhi((unsigned)n, m); }
static inline unsigned div_magic1(int d) { return d == 1 ? 0u : div_magic(d); }
// ds_read_b128 is serviced in the lane groups {0-3,12-15,20-27} and {4-11,16-19,28-31} (per 32-lane half).  Map the 32 rows of
// an MFMA tile to patch pixels so that each group reads 16 CONSECUTIVE pixels (conflict-free 80-byte rows).
__device__ __forceinline__ int row_perm(int l) {
  const bool g1 = (l >= 4 && l < 12) || (l >= 16 && l < 20) || l >= 28;
  if (!g1) return l < 4 ? l : (l < 16 ? l - 8 : l - 12);
  return 16 + (l < 12 ? l - 4 : (l < 20 ? l - 8 : l - 16));
}

struct PrecF16 {
  typedef _Float16 T;
  typedef _Float16 x8 __attribute__((ext_vector_type(8)));
  typedef _Float16 x4 __attribute__((ext_vector_type(4)));
  static constexpr int NPL = 2;
  static constexpr bool SCALED = true;
  static __device__ __forceinline__ void split(float xs, T (&p)[NPL]) {
    p[0] = (T)xs;
    p[1] = (T)(xs - (float)p[0]);
  }
  static __device__ __forceinline__ void split4(f32x4 v, x4 (&p)[NPL]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      T t[NPL];
      split(v[e], t);
      p[0][e] = t[0]; p[1][e] = t[1];
    }
  }
  // product terms (A plane, B plane), smallest first; kernels interleave them over their accumulators
  static constexpr int NTERM = 3;
  static __device__ __forceinline__ constexpr int ta(int t) { return t == 1 ? 1 : 0; }
  static __device__ __forceinline__ constexpr int tb(int t) { return t == 0 ? 1 : 0; }
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0); }
  // ds_read_b64_tr_b16: 4 rows x 16 columns of an LDS image, delivered column-major (cdna_hip_programming.md T10)
  static __device__ __forceinline__ x4 tr_read(const unsigned char* lds) {
    typedef __fp16 v4 __attribute__((vector_size(8)));
    return __builtin_bit_cast(x4, __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) v4*)lds));
  }
};

struct PrecX3 {
  typedef __bf16 T;
  typedef __bf16 x8 __attribute__((ext_vector_type(8)));
  typedef __bf16 x4 __attribute__((ext_vector_type(4)));
  static constexpr int NPL = 3;
  static constexpr bool SCALED = false;
  static __device__ __forceinline__ void split(float x, T (&p)[NPL]) {
    p[0] = (T)x;
    const float r = x - (float)p[0];
    p[1] = (T)r;
    p[2] = (T)(r - (float)p[1]);
  }
#ifdef FS_SPLIT_CVT
  // kernel A/B builds only (measured +-0 against the integer form below, profiles/r04/split_ab.txt): planes by v_cvt_pk_bf16_f32,
  // remainders either by v_dot2c_f32_bf16 with the packed constants (-1, 0) / (0, -1) (FS_SPLIT_CVT=2: 7 VALU per value pair, three
  // wait states per dot) or by unpack + subtract (FS_SPLIT_CVT=1: 11 per pair)
  static __device__ __forceinline__ void split4(f32x4 v, x4 (&p)[NPL]) {
    typedef float f2 __attribute__((ext_vector_type(2)));
    typedef __bf16 b2 __attribute__((ext_vector_type(2)));
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 pk[NPL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      f2 x = {v[2 * h], v[2 * h + 1]};
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) {
        const b2 q = __builtin_convertvector(x, b2);
        const unsigned qu = __builtin_bit_cast(unsigned, q);
        pk[pl][h] = qu;
        if (pl + 1 < NPL) {
#if FS_SPLIT_CVT == 2
          const b2 sel0 = {(__bf16)-1.0f, (__bf16)0.0f}, sel1 = {(__bf16)0.0f, (__bf16)-1.0f};
          const float r0 = __builtin_amdgcn_fdot2_f32_bf16(q, sel0, x[0], false);
          const float r1 = __builtin_amdgcn_fdot2_f32_bf16(q, sel1, x[1], false);
#else
          const float r0 = x[0] - __builtin_bit_cast(float, qu << 16);
          const float r1 = x[1] - __builtin_bit_cast(float, qu & 0xffff0000u);
#endif
          x[0] = r0; x[1] = r1;
        }
      }
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) p[pl] = __builtin_bit_cast(x4, pk[pl]);
  }
#else
  // the same decomposition on 4 values in integer arithmetic: a plane is the fp32 value rounded to its top 16 bits
  // ((bits + 0x8000) & 0xffff0000, ties away from zero), the remainder x - plane is exact in fp32, and the third plane takes
  // what is left (<= 8 significant bits: exact).  x = p0 + p1 + p2 holds exactly, as for split(); the planes can differ from
  // split()'s in the last place on ties.  6 full-rate VALU per value + 1.5 to pack, no conversion instructions.
  static __device__ __forceinline__ void split4(f32x4 v, x4 (&p)[NPL]) {
    typedef unsigned u2 __attribute__((ext_vector_type(2)));
    u2 pk[NPL];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      unsigned hi[2][NPL];
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        float x = v[2 * h + e];
#pragma unroll
        for (int pl = 0; pl < NPL; ++pl) {
          const unsigned b = __builtin_bit_cast(unsigned, x);
          hi[e][pl] = pl + 1 < NPL ? ((b + 0x8000u) & 0xffff0000u) : b;      // last plane: the remainder has <= 8 significant bits
          if (pl + 1 < NPL) x = x - __builtin_bit_cast(float, hi[e][pl]);
        }
      }
#pragma unroll
      for (int pl = 0; pl < NPL; ++pl) pk[pl][h] = __builtin_amdgcn_perm(hi[1][pl], hi[0][pl], 0x07060302u);   // {hi[0]>>16, hi[1]>>16}
    }
#pragma unroll
    for (int pl = 0; pl < NPL; ++pl) p[pl] = __builtin_bit_cast(x4, pk[pl]);
  }
#endif
  static constexpr int NTERM = 6;            // (0,2) (1,1) (2,0) (0,1) (1,0) (0,0)
  static __device__ __forceinline__ constexpr int ta(int t) { return t < 3 ? t : (t == 4 ? 1 : 0); }
  static __device__ __forceinline__ constexpr int tb(int t) { return t < 3 ? 2 - t : (t == 3 ? 1 : 0); }
  static __device__ __forceinline__ f32x16 mfma(x8 a, x8 b, f32x16 c) { return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0); }
  static __device__ __forceinline__ x4 tr_read(const unsigned char* lds) {
    return __builtin_amdgcn_ds_read_tr16_b64_v4bf16((__attribute__((address_space(3))) x4*)lds);
  }
};

}  // namespace fs_split
