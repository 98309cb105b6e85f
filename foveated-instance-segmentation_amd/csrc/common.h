// Shared device helpers for the fovealseg HIP kernels (gfx950 / CDNA4, wave64).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FS_OK 0
#define FS_ERR_ARG 1001   // bad shape / null pointer at the C-ABI boundary

#define FS_LAUNCH_CHECK()                                   \
  do {                                                      \
    hipError_t _e = hipGetLastError();                      \
    if (_e != hipSuccess) return (int)_e;                   \
  } while (0)

#define FS_REQUIRE(cond)          \
  do {                            \
    if (!(cond)) return FS_ERR_ARG; \
  } while (0)

// Kernel-experiment switches (which of two measured variants a shape takes) exist only in builds made with -DFS_EXPERIMENTS
// (`FS_BUILD_EXPERIMENTS=1 python build.py` -> ab/libfovealseg_experiments.so, loaded through FS_HIP_LIB for same-box A/B runs).
// In the shipped library the macro is its default, a compile-time constant: no getenv, and the branch not taken is dead code.
#ifdef FS_EXPERIMENTS
#include <stdlib.h>
#define FS_ENV_INT(name, dflt) ([] { const char* e_ = getenv(name); return e_ ? atoi(e_) : (dflt); }())
#else
#define FS_ENV_INT(name, dflt) (dflt)
#endif

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

// activation codes shared with the host side
enum { FS_ACT_NONE = 0, FS_ACT_RELU = 1, FS_ACT_RELU6 = 2 };

// Dropout keep decision for linear element index e (NHWC order) -- integer hash, restated in numpy by
// oracle/fovealseg_oracle.py:dropout_keep_mask_nhwc for replay tests.  One 32-bit hash (Weyl step + the murmur3 finaliser) serves the
// element PAIR (e >> 1): element 2i takes its low 16 bits, element 2i + 1 its high 16 bits, each compared with the 16-bit threshold
// thresh >> 16 (p = 0.3: 19660 / 65536, 1e-5 from p).  v_mul_lo_u32 is a quarter-rate instruction and the finaliser has two of them:
// the F(2,3) forward epilogue hashes 32 outputs per thread -- 72 such multiplies per tile, a fifth of its epilogue time -- so halving
// the hashes per element is worth having (round 4); a one-multiply finaliser shows 0.2-0.5 % lag and 6 % cross-key correlation of
// the masks and was rejected on that.
__device__ __forceinline__ uint32_t fs_dropout_hash(uint32_t pair, uint32_t key) {
  uint32_t h = pair * 0x9E3779B1u + key;
  h ^= h >> 16;
  h *= 0x85EBCA6Bu;
  h ^= h >> 13;
  h *= 0xC2B2AE35u;
  h ^= h >> 16;
  return h;
}
__device__ __forceinline__ bool fs_dropout_keep(uint32_t e, uint32_t key, uint32_t thresh) {
  const uint32_t h = fs_dropout_hash(e >> 1, key);
  return ((e & 1u) ? (h >> 16) : (h & 0xffffu)) >= (thresh >> 16);
}
// the same for four consecutive elements e .. e + 3 with e EVEN: bit j of the result = fs_dropout_keep(e + j); two hashes
__device__ __forceinline__ uint32_t fs_dropout_keep4(uint32_t e, uint32_t key, uint32_t thresh) {
  const uint32_t t = thresh >> 16;
  const uint32_t h0 = fs_dropout_hash(e >> 1, key), h1 = fs_dropout_hash((e >> 1) + 1u, key);
  return ((h0 & 0xffffu) >= t ? 1u : 0u) | ((h0 >> 16) >= t ? 2u : 0u) | ((h1 & 0xffffu) >= t ? 4u : 0u) | ((h1 >> 16) >= t ? 8u : 0u);
}

__device__ __forceinline__ float fs_act(float v, int act) {
  if (act == FS_ACT_RELU) return v < 0.f ? 0.f : v;      // NaN propagates, as through torch.relu
  if (act == FS_ACT_RELU6) return v < 0.f ? 0.f : (v > 6.f ? 6.f : v);
  return v;
}
// derivative mask from the activation OUTPUT z (ReLU: z>0; ReLU6: 0<z<6, as torch's hardtanh bwd)
__device__ __forceinline__ float fs_act_mask(float z, int act) {
  if (act == FS_ACT_RELU) return z > 0.f ? 1.f : 0.f;
  if (act == FS_ACT_RELU6) return (z > 0.f && z < 6.f) ? 1.f : 0.f;
  return 1.f;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}
__device__ __forceinline__ float wave_min(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fminf(v, __shfl_xor(v, o, 64));
  return v;
}

// Block-wide sum (blockDim.x multiple of 64, <= 1024). `red` = __shared__ scratch of >= 16 T.
template <typename T>
__device__ __forceinline__ T block_sum(T v, T* red) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) red[w] = v;
  __syncthreads();
  T s = 0;
  for (int i = 0; i < nw; ++i) s += red[i];
  return s;
}

static inline int cdiv(long a, long b) { return (int)((a + b - 1) / b); }

// out[i] = (accumulate ? out[i] : 0) + part[0][i] + part[1][i] + ... + part[nslab-1][i], in that order (elementwise.hip).  The second stage
// of every cross-workgroup sum of the library: per-workgroup partials are written with plain stores and added here in index order, so
// no result depends on the order in which workgroups finish.
int fs_slab_reduce(const float* part, int nslab, long n, float* out, int accumulate, hipStream_t stream);
// the same for many slabs of a mid-sized tensor: a first stage folds the slabs into the first 8 IN PLACE (part is scratch)
int fs_slab_reduce_inplace(float* part, int nslab, long n, float* out, int accumulate, hipStream_t stream);
// two sums of equal shape in one launch (a wave per column: meant for n up to a few thousand)
int fs_slab_reduce_pair(const float* part0, const float* part1, int nslab, long n, float* out0, float* out1, int accumulate, hipStream_t stream);
