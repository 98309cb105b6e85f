// Sequence-reduced multi-head attention (head_dim 64, any key count) on the 16-bit matrix pipe in split precision (bf16x3, conv_split.h):
// every fp32 operand -- q*scale, k, v, the softmax probabilities -- enters as three bf16 planes (24 bits, the fp32 operand width) and a
// product is six v_mfma_f32_32x32x16_bf16 with fp32 accumulation.  Against the exact-fp32 kernels of transformer.hip (v_mfma_f32_32x32x2_f32,
// 16 passes per 2 k) that is 3/8 of the matrix-pipe time; the rest of the gain is the TRANSPOSED score tile:
//
//     S^T[key][q] = K (q*scale)^T        A = K rows (keys), B = Q rows (lane = query)
//
// whose accumulator layout puts a query in a LANE and its keys in REGISTERS (element i of lane (r, h) = key (i&3) + 8(i>>2) + 4h, query r).
// The softmax statistics of a query are then one scalar per lane (running max m, running sum l; the two 32-lane halves hold the same
// queries and different keys, one xor-32 shuffle joins them) instead of sixteen 32-lane reductions per tile, and the probabilities are
// ALREADY the B operand of
//
//     O^T[d][q] = V^T P~^T               A = V^T rows (head dim), B = P~^T straight from the registers (k = keys in register order)
//
// so P never goes through LDS.  The k order of that product is the register order: position j of half h in k-step s2 of key tile t is key
// 32t + 16 s2 + 4h + (j&3) + 8(j>>2); the A operand reads V^T[d][those keys] as two 8-byte pieces.
//
// K and V are small (B x Nk x C) and shared by every query tile, so a pre-pass writes them ONCE as planes in the LDS image order
// (fs_attention_split_ws_bytes of scratch): per (batch*head, 64-key chunk) a 48 KB block [K: 3 planes][64 keys][64 d] then
// [V^T: 3 planes][64 d][64 keys], keys beyond Nk zero.  The attention kernel copies a block to LDS per chunk (K rows at a 144-byte pitch:
// conflict-free ds_read_b128 over 16 lanes; V^T rows at 136 bytes: conflict-free ds_read_b64 over 32 lanes) and does no operand splitting
// of its own except P.
//
// Replaces: torch.matmul / softmax / dropout / matmul of transformers==4.46.2 SegformerEfficientSelfAttention.forward (third-party; call
// sites models/segformer.py:9-11,33-37).  Dropout keeps the element hash of transformer.hip (index (bh*N + q)*Nk + key), so a replayed
// mask is the same in both kernel families.
#include "conv_split.h"
#include <cstdlib>

namespace {

using namespace fs_split;
typedef PrecX3 P;
typedef P::x8 X8;
typedef P::x4 X4;

constexpr int HD = 64, KC = 64;
constexpr int KPITCH = 144, VPITCH = 136;                  // bytes per LDS row
constexpr int KPL = KC * KPITCH, VPL = HD * VPITCH;        // bytes per LDS plane
constexpr int LDS_K = 0, LDS_V = 3 * KPL;
constexpr int LDS_BYTES = 3 * KPL + 3 * VPL;               // 53 760: two workgroups per CU
constexpr int BLOCK_BYTES = 6 * KC * HD * 2;               // one packed (bh, chunk) block in HBM: 49 152

__device__ __forceinline__ int acc_row(int i, int h) { return (i & 3) + 8 * (i >> 2) + 4 * h; }
__device__ __forceinline__ X8 cat(X4 lo, X4 hi) { return __builtin_shufflevector(lo, hi, 0, 1, 2, 3, 4, 5, 6, 7); }
__device__ __forceinline__ void split8(f32x4 a, f32x4 b, X8 (&out)[3]) {
  X4 pa[3], pb[3];
  P::split4(a, pa);
  P::split4(b, pb);
#pragma unroll
  for (int pl = 0; pl < 3; ++pl) out[pl] = cat(pa[pl], pb[pl]);
}

// ---- pre-pass: K -> row-major planes, V -> transposed planes, per (bh, chunk) block ----
__global__ __launch_bounds__(256) void attn_pack_kv_kernel(const float* __restrict__ k, const float* __restrict__ v, unsigned char* __restrict__ ws,
                                                           int Nk, int heads, int nchunk) {
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const float* kb = k + (long)b * Nk * C + hd * HD;
  const float* vb = v + (long)b * Nk * C + hd * HD;
  unsigned char* blk = ws + ((long)bh * nchunk + c) * BLOCK_BYTES;
  // item = (key row, channel quad): 64 x 16 items, 4 per thread
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int item = threadIdx.x + 256 * it;
    const int key = item >> 4, c4 = (item & 15) * 4;
    const int gk = c * KC + key;
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (gk < Nk) {
      kv = *reinterpret_cast<const f32x4*>(kb + (long)gk * C + c4);
      vv = *reinterpret_cast<const f32x4*>(vb + (long)gk * C + c4);
    }
    X4 pk[3], pv[3];
    P::split4(kv, pk);
    P::split4(vv, pv);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<X4*>(blk + (pl * KC + key) * (HD * 2) + c4 * 2) = pk[pl];
      __bf16* vt = reinterpret_cast<__bf16*>(blk + 3 * KC * HD * 2 + pl * HD * KC * 2);
#pragma unroll
      for (int e = 0; e < 4; ++e) vt[(c4 + e) * KC + key] = pv[pl][e];
    }
  }
}

// one 48 KB block -> the LDS image (K rows at KPITCH, V^T rows at VPITCH)
__device__ __forceinline__ void stage_block(const unsigned char* __restrict__ blk, unsigned char* lds, int tid) {
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int piece = tid + 256 * it;                 // 3072 pieces of 16 bytes; pieces 0..1535 are K, the rest V^T (uniform per `it`)
    const u32x4 d = *reinterpret_cast<const u32x4*>(blk + piece * 16);
    const int rem = it < 6 ? piece : piece - 1536, pl = rem >> 9, row = (rem >> 3) & 63, seg = rem & 7;
    if (it < 6) {
      *reinterpret_cast<u32x4*>(lds + LDS_K + pl * KPL + row * KPITCH + seg * 16) = d;
    } else {
      typedef unsigned u2 __attribute__((ext_vector_type(2)));
      unsigned char* dst = lds + LDS_V + pl * VPL + row * VPITCH + seg * 16;
      *reinterpret_cast<u2*>(dst) = u2{d.x, d.y};
      *reinterpret_cast<u2*>(dst + 8) = u2{d.z, d.w};
    }
  }
}

__global__ __launch_bounds__(256, 2) void attn_split_fwd_kernel(const float* __restrict__ q, const unsigned char* __restrict__ ws,
                                                             float* __restrict__ o, float* __restrict__ lse, unsigned* __restrict__ mask,
                                                             int N, int Nk, int heads, float scale, float drop_scale, uint32_t thresh,
                                                             uint32_t key) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const int qrow = blockIdx.x * 128 + wave * 32 + r;           // this lane's query
  const bool qok = qrow < N;
  const int nchunk = (Nk + KC - 1) / KC;
  const unsigned char* blk0 = ws + (long)bh * nchunk * BLOCK_BYTES;

  // B operand of S^T: (q * scale * log2 e)[qrow][16 ks + 8 h .. + 7] as planes: the scores come out in base-2 units, so every
  // exponential below is one v_exp_f32 (the softmax is VALU-bound here: 32 probabilities per lane and chunk against 96 MFMAs)
  X8 qf[4][3];
  {
    const float* qp = q + ((long)b * N + (qok ? qrow : 0)) * C + hd * HD + 8 * h;
    const float sc2 = scale * 1.44269504088896340736f;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, c2 = {0.f, 0.f, 0.f, 0.f};
      if (qok) { a = *reinterpret_cast<const f32x4*>(qp + 16 * ks) * sc2; c2 = *reinterpret_cast<const f32x4*>(qp + 16 * ks + 4) * sc2; }
      split8(a, c2, qf[ks]);
    }
  }
  f32x16 O[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { O[0][i] = 0.f; O[1][i] = 0.f; }
  float m = -INFINITY, l = 0.f;
  const uint32_t ebase = (uint32_t)(((long)bh * N + qrow) * Nk);

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();                           // every wave is done with the previous chunk's image
    stage_block(blk0 + (long)c * BLOCK_BYTES, lds, tid);
    __syncthreads();
    f32x16 S[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
#pragma unroll
      for (int i = 0; i < 16; ++i) S[t][i] = 0.f;
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        X8 kf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl)
          kf[pl] = *reinterpret_cast<const X8*>(lds + LDS_K + pl * KPL + (32 * t + r) * KPITCH + (16 * ks + 8 * h) * 2);
#pragma unroll
        for (int tm = 0; tm < P::NTERM; ++tm) S[t] = P::mfma(kf[P::ta(tm)], qf[ks][P::tb(tm)], S[t]);
      }
    }
    // online softmax, per lane (= per query); keys beyond Nk are excluded
    float mloc = -INFINITY;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const bool ok = c * KC + 32 * t + acc_row(i, h) < Nk;
        S[t][i] = ok ? S[t][i] : -INFINITY;
        mloc = fmaxf(mloc, S[t][i]);
      }
    mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
    const float mn = fmaxf(m, mloc);
    const float alpha = __builtin_amdgcn_exp2f(m - mn);
    m = mn;
    float psum = 0.f;
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      unsigned bits = 0u;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        float p = __builtin_amdgcn_exp2f(S[t][i] - mn);            // 2^(-inf) = 0 for the excluded keys
        psum += p;
        if (thresh != 0u) {
          const bool keep = fs_dropout_keep(ebase + (uint32_t)(c * KC + 32 * t + acc_row(i, h)), key, thresh);
          p = keep ? p * drop_scale : 0.f;
          bits |= keep ? (1u << acc_row(i, h)) : 0u;
        }
        S[t][i] = p;
      }
      // the keep decisions of (query, keys 64 c + 32 t .. + 31) as one word for the backward kernels, which then never hash
      if (thresh != 0u && mask != nullptr) {
        bits |= __shfl_xor(bits, 32, 64);
        const int nword = (Nk + 31) >> 5;
        if (qok && h == t && 2 * c + t < nword) mask[((long)bh * N + qrow) * nword + 2 * c + t] = bits;
      }
    }
    l = l * alpha + psum;
#pragma unroll
    for (int i = 0; i < 16; ++i) { O[0][i] *= alpha; O[1][i] *= alpha; }
    // O^T += V^T P~^T, k-steps in register order
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        X8 pf[3];
        split8(f32x4{S[t][8 * s2], S[t][8 * s2 + 1], S[t][8 * s2 + 2], S[t][8 * s2 + 3]},
               f32x4{S[t][8 * s2 + 4], S[t][8 * s2 + 5], S[t][8 * s2 + 6], S[t][8 * s2 + 7]}, pf);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          X8 vf[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const unsigned char* a0 = lds + LDS_V + pl * VPL + (32 * dt + r) * VPITCH + (32 * t + 16 * s2 + 4 * h) * 2;
            vf[pl] = cat(*reinterpret_cast<const X4*>(a0), *reinterpret_cast<const X4*>(a0 + 16));
          }
#pragma unroll
          for (int tm = 0; tm < P::NTERM; ++tm) O[dt] = P::mfma(vf[P::ta(tm)], pf[P::tb(tm)], O[dt]);
        }
      }
  }
  l += __shfl_xor(l, 32, 64);
  if (!qok) return;
  const float inv = 1.f / l;
  float* orow = o + ((long)b * N + qrow) * C + hd * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(orow + 32 * dt + 8 * g + 4 * h) =
          f32x4{O[dt][4 * g], O[dt][4 * g + 1], O[dt][4 * g + 2], O[dt][4 * g + 3]} * inv;
  if (h == 0) lse[(long)bh * N + qrow] = (m + log2f(l)) * 0.69314718055994530942f;      // natural log-sum-exp of the scores
}

// ---- backward, dQ: the same transposed tiles with the query on the lane -----------------------------------------------------------
//     S^T = K (q*scale)^T,  dP~^T = V dO^T   (A = K / V rows of a key tile, B = the lane's q / dO row as planes, held in registers)
//     dS^T = P^T o (mask * dP~^T * drop_scale - D)     per lane: lse and D of the lane's query are scalars
//     dQ^T[d][q] += K^T dS^T                 (A = the K image read TRANSPOSED, keys in register order; B = dS^T planes from the registers)
// No atomics (a workgroup owns its 128 queries), no LDS round trip for dS.  Packed per (bh, 64-key chunk): [K][V] row-major planes, 48 KB,
// one LDS image each at a 144-byte pitch; the two 32-key tiles of a chunk are processed one after the other between the same pair of
// barriers (the time of these kernels follows the number of barrier rounds: 32-key chunks with a separate K^T image took 171 us per stage-3
// launch, this form [see DESIGN.md 4b]).
constexpr int BW_BLOCK = 6 * KC * HD * 2;                  // K rm, V rm: 49 152
constexpr int L2_K = 0, L2_V = 3 * KPL;
constexpr int LDS2_BYTES = 6 * KPL;                        // 55 296

__global__ __launch_bounds__(256) void attn_pack_bwd_kernel(const float* __restrict__ k, const float* __restrict__ v, unsigned char* __restrict__ ws,
                                                            int Nk, int heads, int nchunk) {
  const int c = blockIdx.x, bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const float* kb = k + (long)b * Nk * C + hd * HD;
  const float* vb = v + (long)b * Nk * C + hd * HD;
  unsigned char* blk = ws + ((long)bh * nchunk + c) * BW_BLOCK;
#pragma unroll
  for (int it = 0; it < 4; ++it) {
    const int item = threadIdx.x + 256 * it;       // (key row, channel quad): 64 x 16
    const int key = item >> 4, c4 = (item & 15) * 4;
    const int gk = c * KC + key;
    f32x4 kv = {0.f, 0.f, 0.f, 0.f}, vv = {0.f, 0.f, 0.f, 0.f};
    if (gk < Nk) {
      kv = *reinterpret_cast<const f32x4*>(kb + (long)gk * C + c4);
      vv = *reinterpret_cast<const f32x4*>(vb + (long)gk * C + c4);
    }
    X4 pk[3], pv[3];
    P::split4(kv, pk);
    P::split4(vv, pv);
#pragma unroll
    for (int pl = 0; pl < 3; ++pl) {
      *reinterpret_cast<X4*>(blk + (pl * KC + key) * (HD * 2) + c4 * 2) = pk[pl];
      *reinterpret_cast<X4*>(blk + 3 * KC * HD * 2 + (pl * KC + key) * (HD * 2) + c4 * 2) = pv[pl];
    }
  }
}

__device__ __forceinline__ void stage_block_bwd(const unsigned char* __restrict__ blk, unsigned char* lds, int tid) {
#pragma unroll
  for (int it = 0; it < 12; ++it) {
    const int piece = tid + 256 * it;               // 3072 pieces of 16 bytes: 1536 K, 1536 V (uniform per `it`)
    const u32x4 d = *reinterpret_cast<const u32x4*>(blk + piece * 16);
    const int rem = it < 6 ? piece : piece - 1536, pl = rem >> 9, row = (rem >> 3) & 63, seg = rem & 7;
    *reinterpret_cast<u32x4*>(lds + (it < 6 ? L2_K : L2_V) + pl * KPL + row * KPITCH + seg * 16) = d;
  }
}

__global__ __launch_bounds__(256, 2) void attn_split_bwd_dq_kernel(const float* __restrict__ q, const float* __restrict__ go,
                                                                const unsigned char* __restrict__ ws, const float* __restrict__ lse,
                                                                const float* __restrict__ D, const unsigned* __restrict__ mask,
                                                                float* __restrict__ dq, int N, int Nk, int heads, float scale,
                                                                float drop_scale, uint32_t thresh, uint32_t key) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS2_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const int qrow = blockIdx.x * 128 + wave * 32 + r;
  const bool qok = qrow < N;
  const int nchunk = (Nk + KC - 1) / KC;
  const unsigned char* blk0 = ws + (long)bh * nchunk * BW_BLOCK;

  const float L2e = 1.44269504088896340736f;
  X8 qf[4][3], gf[4][3];          // B operands: (q*scale*log2 e) and dO rows of this lane's query
  {
    const long off = ((long)b * N + (qok ? qrow : 0)) * C + hd * HD + 8 * h;
    const float sc2 = scale * L2e;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, a2 = a, g = a, g2 = a;
      if (qok) {
        a = *reinterpret_cast<const f32x4*>(q + off + 16 * ks) * sc2; a2 = *reinterpret_cast<const f32x4*>(q + off + 16 * ks + 4) * sc2;
        g = *reinterpret_cast<const f32x4*>(go + off + 16 * ks); g2 = *reinterpret_cast<const f32x4*>(go + off + 16 * ks + 4);
      }
      split8(a, a2, qf[ks]);
      split8(g, g2, gf[ks]);
    }
  }
  const float Lb = (qok ? lse[(long)bh * N + qrow] : INFINITY) * L2e;       // -Lb starts the score chain: p = 2^S, 0 for lanes past the end
  const float Dq = qok ? D[(long)bh * N + qrow] : 0.f;
  const int nword = (Nk + 31) >> 5;                 // the forward's keep words of this query (key tile = word), when it left them
  const unsigned* mrow = mask != nullptr && qok && thresh != 0u ? mask + ((long)bh * N + qrow) * nword : nullptr;
  unsigned mword[2] = {mrow != nullptr ? mrow[0] : 0u, mrow != nullptr && nword > 1 ? mrow[1] : 0u};
  const uint32_t ebase = (uint32_t)(((long)bh * N + qrow) * Nk);
  f32x16 dQ[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { dQ[0][i] = 0.f; dQ[1][i] = 0.f; }
  // transposed-read lane constants (conv_wgrad.hip): lane 4 q' + p of a 16-lane group addresses row q', columns 4 p .. 4 p + 3
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, cb = 16 * ((lane >> 4) & 1);
  const int tr_ofs = tq * KPITCH + (cb + 4 * tp) * 2;

  for (int c = 0; c < nchunk; ++c) {
    __syncthreads();
    stage_block_bwd(blk0 + (long)c * BW_BLOCK, lds, tid);
    __syncthreads();
    const unsigned mw0 = mword[0], mw1 = mword[1];
    if (mrow != nullptr) {                        // (for the next chunk: in flight behind this one's products)
      mword[0] = 2 * c + 2 < nword ? mrow[2 * c + 2] : 0u;
      mword[1] = 2 * c + 3 < nword ? mrow[2 * c + 3] : 0u;
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      f32x16 S, dP;
#pragma unroll
      for (int i = 0; i < 16; ++i) { S[i] = -Lb; dP[i] = 0.f; }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        X8 kf[3], vf[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          kf[pl] = *reinterpret_cast<const X8*>(lds + L2_K + pl * KPL + (32 * t + r) * KPITCH + (16 * ks + 8 * h) * 2);
          vf[pl] = *reinterpret_cast<const X8*>(lds + L2_V + pl * KPL + (32 * t + r) * KPITCH + (16 * ks + 8 * h) * 2);
        }
#pragma unroll
        for (int tm = 0; tm < P::NTERM; ++tm) {
          S = P::mfma(kf[P::ta(tm)], qf[ks][P::tb(tm)], S);
          dP = P::mfma(vf[P::ta(tm)], gf[ks][P::tb(tm)], dP);
        }
      }
      const unsigned mw = t == 0 ? mw0 : mw1;
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int keyi = c * KC + 32 * t + acc_row(i, h);
        const float p = keyi < Nk ? __builtin_amdgcn_exp2f(S[i]) : 0.f;
        float mk = 1.f;
        if (thresh != 0u) {
          const bool keep = mask != nullptr ? ((mw >> acc_row(i, h)) & 1u) != 0u : fs_dropout_keep(ebase + (uint32_t)keyi, key, thresh);
          mk = keep ? drop_scale : 0.f;
        }
        S[i] = p * (mk * dP[i] - Dq);
      }
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        X8 df[3];
        split8(f32x4{S[8 * s2], S[8 * s2 + 1], S[8 * s2 + 2], S[8 * s2 + 3]}, f32x4{S[8 * s2 + 4], S[8 * s2 + 5], S[8 * s2 + 6], S[8 * s2 + 7]}, df);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
          X8 tf[3];
#pragma unroll
          for (int pl = 0; pl < 3; ++pl) {
            const unsigned char* a0 = lds + L2_K + pl * KPL + (32 * t + 16 * s2 + 4 * h) * KPITCH + 64 * dt + tr_ofs;
            tf[pl] = cat(P::tr_read(a0), P::tr_read(a0 + 8 * KPITCH));
          }
#pragma unroll
          for (int tm = 0; tm < P::NTERM; ++tm) dQ[dt] = P::mfma(tf[P::ta(tm)], df[P::tb(tm)], dQ[dt]);
        }
      }
    }
  }
  if (!qok) return;
  float* drow = dq + ((long)b * N + qrow) * C + hd * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(drow + 32 * dt + 8 * g + 4 * h) =
          f32x4{dQ[dt][4 * g], dQ[dt][4 * g + 1], dQ[dt][4 * g + 2], dQ[dt][4 * g + 3]} * scale;
}

// ---- backward, dK and dV: the key on the lane -------------------------------------------------------------------------------------
//     S[q][key] = (q*scale) K^T,  dP~[q][key] = dO V^T      A = rows of the Q / dO slice image, B = the lane's K / V row as planes (registers)
//     P~ = dropout(exp(S - lse[q])),  dS = P o (mask * dP~ * drop_scale - D[q])     (q is the REGISTER index: lse and D come from LDS)
//     dV^T[d][key] += dO^T P~,   dK^T[d][key] += (q*scale)^T dS      A = the slice image read TRANSPOSED (ds_read_b64_tr_b16, q in register
//                                                                     order), B = P~ / dS planes straight from the registers
// A wave owns one 32-key tile for the whole sweep over its query range (accumulators in registers, no sum across waves); a workgroup = four
// tiles sharing the 32-query slice images [Q*scale][dO]: 3 planes x 32 rows x 144 B each, written once (split in the kernel) and read by
// rows for S / dP and transposed for the two output products.  Holding K and V planes (96 registers) plus both accumulators (64) plus
// the S and dP tiles does not fit 256 registers without spills, so the two outputs are two instantiations: DK = false computes S -> dV,
// DK = true computes S, dP -> dK (5 tile products per (slice, tile) where a fused kernel would need 4).
// (Measured and rejected: ONE launch of wave pairs sharing a key tile -- wave A: S -> p -> P~ -> dV^T, wave B: dP -> dS -> dK^T, p handed over
// through LDS; 48 MFMAs per wave and slice, 192 registers, no S recompute -- 497 us against 439 us: the time of these kernels follows the number
// of (wave, slice) rounds, not their MFMA count, and the pair form has as many rounds plus a third barrier.)
constexpr int QS = 32;                                     // queries per slice
constexpr int SPL = QS * KPITCH;                           // one plane of a slice image: 4 608
constexpr int L3_Q = 0, L3_G = 3 * SPL, L3_L = 6 * SPL;    // images, then lse[32] (base-2 units), D[32], keep words [4 key tiles][32]
constexpr int LDS3_BYTES = 6 * SPL + 6 * QS * 4;           // 28 416

template <bool DK>
__global__ __launch_bounds__(256, 2) void attn_split_bwd_kv_kernel(const float* __restrict__ q, const float* __restrict__ k,
                                                                const float* __restrict__ v, const float* __restrict__ go,
                                                                const float* __restrict__ lse, const float* __restrict__ D,
                                                                const unsigned* __restrict__ mask, float* __restrict__ dout, int N, int Nk,
                                                                int heads, float scale,
                                                                float drop_scale, uint32_t thresh, uint32_t key, int slices_per_split,
                                                                long part_stride) {
  __shared__ __attribute__((aligned(16))) unsigned char lds[LDS3_BYTES];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r = lane & 31, h = lane >> 5;
  const int bh = blockIdx.y, b = bh / heads, hd = bh - b * heads;
  const int C = heads * HD;
  const int krow = (blockIdx.x * 4 + wave) * 32 + r;          // this lane's key
  const bool kok = krow < Nk;
  const float* qb = q + (long)b * N * C + hd * HD;
  const float* gb = go + (long)b * N * C + hd * HD;

  // B operands: the lane's K row (and V row for dK) as planes, [16 ks + 8 h .. + 7]
  X8 kf[4][3], vf[DK ? 4 : 1][3];
  {
    const long off = ((long)b * Nk + (kok ? krow : 0)) * C + hd * HD + 8 * h;
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      f32x4 a = {0.f, 0.f, 0.f, 0.f}, a2 = a, g = a, g2 = a;
      if (kok) {
        a = *reinterpret_cast<const f32x4*>(k + off + 16 * ks); a2 = *reinterpret_cast<const f32x4*>(k + off + 16 * ks + 4);
        if (DK) { g = *reinterpret_cast<const f32x4*>(v + off + 16 * ks); g2 = *reinterpret_cast<const f32x4*>(v + off + 16 * ks + 4); }
      }
      split8(a, a2, kf[ks]);
      if (DK) split8(g, g2, vf[ks]);
    }
  }
  f32x16 acc[2];
#pragma unroll
  for (int i = 0; i < 16; ++i) { acc[0][i] = 0.f; acc[1][i] = 0.f; }

  // staging: item = (row tid >> 4 (+16), channel quad tid & 15) of the slice, Q*scale and dO
  const int srow = tid >> 4, c4 = (tid & 15) * 4;
  // transposed-read lane constants (conv_wgrad.hip): lane 4 q' + p of a 16-lane group addresses row q', columns 4 p .. 4 p + 3 of the
  // group's 16 columns cb .. cb + 15; the lane then holds column cb + (lane & 15) of the four rows
  const int i16 = lane & 15, tq = i16 >> 2, tp = i16 & 3, cb = 16 * ((lane >> 4) & 1);
  const int tr_ofs = tq * KPITCH + (cb + 4 * tp) * 2;          // + row base * KPITCH + 64 dt + plane
  const int nslice = (N + QS - 1) / QS;
  const int s_begin = blockIdx.z * slices_per_split;
  const int s_end = s_begin + slices_per_split < nslice ? s_begin + slices_per_split : nslice;
  const float* lsel = reinterpret_cast<const float*>(lds + L3_L);

  // the next slice's rows travel from HBM while the current slice is computed: two Q and two dO float4s and one lse / D word per thread
  f32x4 nq[2], ng[2];
  float nl = 0.f;
  const int nword = (Nk + 31) >> 5;
  auto fetch = [&](int sl_) {
    const int q0_ = sl_ * QS;
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int qr = q0_ + srow + 16 * it;
      nq[it] = f32x4{0.f, 0.f, 0.f, 0.f}; ng[it] = nq[it];
      if (qr < N) {
        nq[it] = *reinterpret_cast<const f32x4*>(qb + (long)qr * C + c4);
        ng[it] = *reinterpret_cast<const f32x4*>(gb + (long)qr * C + c4);
      }
    }
    if (tid < 2 * QS) {                    // lse (+inf past the end: p = 0) and D of the slice's queries
      const int qr = q0_ + (tid & 31);
      const bool ok = qr < N;
      nl = tid < QS ? (ok ? lse[(long)bh * N + qr] * 1.44269504088896340736f : INFINITY) : (ok ? D[(long)bh * N + qr] : 0.f);
    } else if (tid < 6 * QS && mask != nullptr) {      // the forward's keep words: (query, this workgroup's key tile (tid >> 5) - 2)
      const int qr = q0_ + (tid & 31), tile = blockIdx.x * 4 + (tid >> 5) - 2;
      nl = 0.f;
      if (qr < N && tile < nword) nl = __builtin_bit_cast(float, mask[((long)bh * N + qr) * nword + tile]);
    }
  };
  if (s_begin < s_end) fetch(s_begin);

  for (int sl = s_begin; sl < s_end; ++sl) {
    const int q0 = sl * QS;
    __syncthreads();                       // every wave is done with the previous slice
#pragma unroll
    for (int it = 0; it < 2; ++it) {
      const int row = srow + 16 * it;
      X4 pa[3], pg[3];
      P::split4(nq[it] * (scale * 1.44269504088896340736f), pa);      // base-2 units: dK = acc * ln 2 at the end
      P::split4(ng[it], pg);
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        *reinterpret_cast<X4*>(lds + L3_Q + pl * SPL + row * KPITCH + c4 * 2) = pa[pl];
        *reinterpret_cast<X4*>(lds + L3_G + pl * SPL + row * KPITCH + c4 * 2) = pg[pl];
      }
    }
    if (tid < 6 * QS) reinterpret_cast<float*>(lds + L3_L)[tid] = nl;
    __syncthreads();
    if (sl + 1 < s_end) fetch(sl + 1);

    // the per-query constants of this lane's 16 score rows, requested from LDS BEFORE the product chains so that their latency hides
    // behind the MFMAs (read at their uses they were 48 exposed LDS round trips per slice); -lse is the chain's initial accumulator
    f32x16 S, dP, Dv;
    unsigned mwv[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) { S[i] = -lsel[acc_row(i, h)]; dP[i] = 0.f; }
    auto row_consts = [&]() {       // one batch of LDS reads: D[q] and the keep word of each of the lane's 16 score rows
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int ql = acc_row(i, h);
        Dv[i] = DK ? lsel[QS + ql] : 0.f;
        mwv[i] = mask != nullptr ? __builtin_bit_cast(unsigned, lsel[(2 + wave) * QS + ql]) : 0u;
      }
    };
    if (!DK) row_consts();          // (the dK instantiation has no registers to spare during the chains: it batches them behind)
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      X8 qa[3], ga[3];
#pragma unroll
      for (int pl = 0; pl < 3; ++pl) {
        qa[pl] = *reinterpret_cast<const X8*>(lds + L3_Q + pl * SPL + r * KPITCH + (16 * ks + 8 * h) * 2);
        if (DK) ga[pl] = *reinterpret_cast<const X8*>(lds + L3_G + pl * SPL + r * KPITCH + (16 * ks + 8 * h) * 2);
      }
#pragma unroll
      for (int tm = 0; tm < P::NTERM; ++tm) {
        S = P::mfma(qa[P::ta(tm)], kf[ks][P::tb(tm)], S);
        if (DK) dP = P::mfma(ga[P::ta(tm)], vf[ks][P::tb(tm)], dP);
      }
    }
    if (DK) row_consts();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int ql = acc_row(i, h);
      const float p = kok ? __builtin_amdgcn_exp2f(S[i]) : 0.f;          // S = (q k) log2 e - lse log2 e
      float mk = 1.f;
      if (thresh != 0u) {
        const bool keep = mask != nullptr ? ((mwv[i] >> r) & 1u) != 0u
                                          : fs_dropout_keep((uint32_t)(((long)bh * N + q0 + ql) * Nk) + (uint32_t)krow, key, thresh);
        mk = keep ? drop_scale : 0.f;
      }
      S[i] = DK ? p * (mk * dP[i] - Dv[i]) : p * mk;
    }
    // output product: A = the dO (dV) or Q*scale (dK) image read transposed, rows d, k = queries in register order
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      X8 bf[3];
      split8(f32x4{S[8 * s2], S[8 * s2 + 1], S[8 * s2 + 2], S[8 * s2 + 3]}, f32x4{S[8 * s2 + 4], S[8 * s2 + 5], S[8 * s2 + 6], S[8 * s2 + 7]}, bf);
#pragma unroll
      for (int dt = 0; dt < 2; ++dt) {
        X8 af[3];
#pragma unroll
        for (int pl = 0; pl < 3; ++pl) {
          const unsigned char* a0 = lds + (DK ? L3_Q : L3_G) + pl * SPL + (16 * s2 + 4 * h) * KPITCH + 64 * dt + tr_ofs;
          af[pl] = cat(P::tr_read(a0), P::tr_read(a0 + 8 * KPITCH));
        }
#pragma unroll
        for (int tm = 0; tm < P::NTERM; ++tm) acc[dt] = P::mfma(af[P::ta(tm)], bf[P::tb(tm)], acc[dt]);
      }
    }
  }
  if (!kok) return;                  // (a split always has slices; its accumulators are zero otherwise, and every split writes its rows)
  if (DK) {
#pragma unroll
    for (int i = 0; i < 16; ++i) { acc[0][i] *= 0.69314718055994530942f; acc[1][i] *= 0.69314718055994530942f; }
  }
  // one split: the gradient itself; several: this split's partial tensor (summed by attn_sum_parts_kernel -- plain 16-byte stores where
  // fp32 atomics scattered over 64 rows per instruction cost more than the whole sweep)
  float* drow = dout + (long)blockIdx.z * part_stride + ((long)b * Nk + krow) * C + hd * HD;
#pragma unroll
  for (int dt = 0; dt < 2; ++dt)
#pragma unroll
    for (int g = 0; g < 4; ++g)
      *reinterpret_cast<f32x4*>(drow + 32 * dt + 8 * g + 4 * h) = f32x4{acc[dt][4 * g], acc[dt][4 * g + 1], acc[dt][4 * g + 2], acc[dt][4 * g + 3]};
}

// dk / dv = sum over the query splits of their partial tensors (rows a split did not reach are never read: every split writes every key row)
__global__ __launch_bounds__(256) void attn_sum_parts_kernel(const float* __restrict__ parts, float* __restrict__ dk, float* __restrict__ dv,
                                                             long n4, int nsplit) {
  const long i = (long)blockIdx.x * 256 + threadIdx.x;
  if (i >= 2 * n4) return;
  const bool second = i >= n4;
  const long j = second ? i - n4 : i;
  const f32x4* src = reinterpret_cast<const f32x4*>(parts) + (second ? (long)nsplit * n4 : 0) + j;
  f32x4 a = src[0];
  for (int s_ = 1; s_ < nsplit; ++s_) a += src[(long)s_ * n4];
  reinterpret_cast<f32x4*>(second ? dv : dk)[j] = a;
}

}  // namespace

extern "C" {

// include/fovealseg.h: words of the dropout keep mask the forward can leave for the backward: B*heads*N rows of ceil(Nk / 32)
long fs_attention_mask_words(int B, int N, int Nk, int heads) {
  if (B <= 0 || N <= 0 || Nk <= 0 || heads <= 0) return 0;
  return (long)B * heads * N * ((Nk + 31) >> 5);
}

// include/fovealseg.h: scratch of the split-precision attention entry points (the packed K / V^T planes)
long fs_attention_split_ws_bytes(int B, int Nk, int heads) {
  if (B <= 0 || Nk <= 0 || heads <= 0) return 0;
  return (long)B * heads * ((Nk + KC - 1) / KC) * BLOCK_BYTES;
}

int fs_attention_fwd_split(const float* q, const float* k, const float* v, float* o, float* lse, unsigned* mask, void* ws, long ws_bytes, int B,
                           int N, int Nk, int heads, float scale, float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && o && lse && ws && B > 0 && N > 0 && Nk > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f);
  FS_REQUIRE((long)B * heads * N * Nk < 4294967296L && (long)B * heads < 65536 && ws_bytes >= fs_attention_split_ws_bytes(B, Nk, heads));
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  const int nchunk = (Nk + KC - 1) / KC;
  hipLaunchKernelGGL(attn_pack_kv_kernel, dim3(nchunk, B * heads), dim3(256), 0, stream, k, v, reinterpret_cast<unsigned char*>(ws), Nk, heads, nchunk);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_split_fwd_kernel, dim3((N + 127) / 128, B * heads), dim3(256), 0, stream, q, reinterpret_cast<const unsigned char*>(ws), o,
                     lse, mask, N, Nk, heads, scale, ds, thresh, key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// include/fovealseg.h: scratch of fs_attention_bwd_dq_split (K and V planes per 64-key chunk)
static long attn_dq_pack_bytes(int B, int Nk, int heads) { return (long)B * heads * ((Nk + KC - 1) / KC) * BW_BLOCK; }
long fs_attention_bwd_split_ws_bytes(int B, int Nk, int heads) {
  if (B <= 0 || Nk <= 0 || heads <= 0) return 0;
  return attn_dq_pack_bytes(B, Nk, heads) + 2L * 8 * B * Nk * heads * HD * 4;      // + the dK / dV partial tensors of up to 8 query splits
}
// offset of the partial tensors inside that scratch (fs_attention_bwd_split hands them to fs_attention_bwd_dkv_split)
long fs_attention_bwd_split_parts_offset(int B, int Nk, int heads) { return attn_dq_pack_bytes(B, Nk, heads); }

// dQ of the attention backward in split precision; D = B*heads*N floats holding rowsum(dO * O) (fs_attention_bwd computes them into its scratch)
int fs_attention_bwd_dq_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                              const unsigned* mask, float* dq, void* ws, long ws_bytes, int B, int N, int Nk, int heads, float scale,
                              float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && go && lse && D && dq && ws && B > 0 && N > 0 && Nk > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f);
  FS_REQUIRE((long)B * heads * N * Nk < 4294967296L && (long)B * heads < 65536 && ws_bytes >= attn_dq_pack_bytes(B, Nk, heads));
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  const int nchunk = (Nk + KC - 1) / KC;
  hipLaunchKernelGGL(attn_pack_bwd_kernel, dim3(nchunk, B * heads), dim3(256), 0, stream, k, v, reinterpret_cast<unsigned char*>(ws), Nk, heads, nchunk);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_split_bwd_dq_kernel, dim3((N + 127) / 128, B * heads), dim3(256), 0, stream, q, go,
                     reinterpret_cast<const unsigned char*>(ws), lse, D, mask, dq, N, Nk, heads, scale, ds, thresh, key);
  FS_LAUNCH_CHECK();
  return FS_OK;
}

// split count of the dK / dV sweep over the query range: 512 workgroups are resident at once (two per CU), the launch runs in
// ceil(workgroups / 512) rounds of ceil(nslice / nsplit) slices; fewest slice-times wins, a split costing a quarter slice (its K / V
// prologue and its partial tensor).  Stage 3 of configs[3] (320 workgroups per split): 3 splits = 2 rounds x 17 slices instead of 1 x 50.
static int attn_kv_splits(int B, int N, int Nk, int heads) {
  const long base = (long)((Nk + 127) / 128) * B * heads;
  const int nslice = (N + QS - 1) / QS;
  int nsplit = 1;
  double best = 1e30;
  for (int c = 1; c <= 8 && c <= nslice; ++c) {
    const long rounds = (base * c + 511) / 512;
    const double cost = (double)rounds * ((nslice + c - 1) / c) + 0.25 * c;
    if (cost < best) { best = cost; nsplit = c; }
  }
  const int sps = (nslice + nsplit - 1) / nsplit;
  return (nslice + sps - 1) / sps;
}

// dK and dV of the attention backward in split precision.  D = rowsum(dO * O).  dk / dv are overwritten.  parts (nullable) = 2 * 8 * B * Nk *
// heads * 64 floats of scratch for the per-split partial tensors; without it the query range is not split.
int fs_attention_bwd_dkv_split(const float* q, const float* k, const float* v, const float* go, const float* lse, const float* D,
                               const unsigned* mask, float* dk, float* dv, float* parts, int B, int N, int Nk, int heads, float scale,
                               float drop_p, uint32_t key, hipStream_t stream) {
  FS_REQUIRE(q && k && v && go && lse && D && dk && dv && B > 0 && N > 0 && Nk > 0 && heads > 0 && drop_p >= 0.f && drop_p < 1.f);
  FS_REQUIRE((long)B * heads * N * Nk < 4294967296L && (long)B * heads < 65536);
  float ds = 1.f; uint32_t thresh = 0u;
  if (drop_p > 0.f) { ds = 1.0f / (float)(1.0 - (double)drop_p); thresh = (uint32_t)((double)drop_p * 4294967296.0); }
  const int nkb = (Nk + 127) / 128, nslice = (N + QS - 1) / QS;
  const int nsplit = parts != nullptr ? attn_kv_splits(B, N, Nk, heads) : 1;
  const int sps = (nslice + nsplit - 1) / nsplit;
  const long elems = (long)B * Nk * heads * HD;
  const dim3 grid(nkb, B * heads, nsplit);
  float* out_v = nsplit > 1 ? parts + (long)nsplit * elems : dv;
  float* out_k = nsplit > 1 ? parts : dk;
  hipLaunchKernelGGL(attn_split_bwd_kv_kernel<false>, grid, dim3(256), 0, stream, q, k, v, go, lse, D, mask, out_v, N, Nk, heads, scale, ds, thresh,
                     key, sps, nsplit > 1 ? elems : 0L);
  FS_LAUNCH_CHECK();
  hipLaunchKernelGGL(attn_split_bwd_kv_kernel<true>, grid, dim3(256), 0, stream, q, k, v, go, lse, D, mask, out_k, N, Nk, heads, scale, ds, thresh,
                     key, sps, nsplit > 1 ? elems : 0L);
  FS_LAUNCH_CHECK();
  if (nsplit > 1) {
    const long n4 = elems / 4;
    hipLaunchKernelGGL(attn_sum_parts_kernel, dim3((unsigned)((2 * n4 + 255) / 256)), dim3(256), 0, stream, parts, dk, dv, n4, nsplit);
    FS_LAUNCH_CHECK();
  }
  return FS_OK;
}

}  // extern "C"
