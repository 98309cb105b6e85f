// Cycles per v_mfma_f32_32x32x16_bf16 when a wave issues chains of 6 MFMAs into ONE accumulator (the 3x3 bwd-weight class kernel:
// six product terms of a tap into acc[tap]) against the same MFMAs alternating between two accumulators, with one and with two waves
// per SIMD.  Registers only.  hipcc -O3 --offload-arch=gfx950 tools/probes/mfma_dep_probe.hip -o /tmp/mfma_dep && /tmp/mfma_dep
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <vector>
typedef __bf16 bf8 __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE>      // 0: chains of 6 into one accumulator, then the next; 1: two accumulators alternate; 2: four accumulators rotate
__global__ __launch_bounds__(512) void probe(float* out, long long* cyc, int iters) {
  const int lane = threadIdx.x & 63;
  bf8 a[3], b[3];
  for (int p = 0; p < 3; ++p)
    for (int e = 0; e < 8; ++e) { a[p][e] = (__bf16)(0.001f * ((lane * 7 + p * 3 + e) % 13)); b[p][e] = (__bf16)(0.002f * ((lane * 5 + p + e * 3) % 11)); }
  f32x16 acc[4] = {};
  __syncthreads();
  const long long t0 = clock64();
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
#pragma unroll
      for (int t = 0; t < 6; ++t) {
        const int ta = t < 3 ? t : (t == 4 ? 1 : 0), tb = t < 3 ? 2 - t : (t == 3 ? 1 : 0);
        const int k = MODE == 0 ? g : (MODE == 1 ? ((g & 2) | (t & 1)) : ((t + g) & 3));
        acc[k] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[ta], b[tb], acc[k], 0, 0, 0);
      }
    }
  }
  const long long t1 = clock64();
  float s = 0.f;
  for (int k = 0; k < 4; ++k) for (int r = 0; r < 16; ++r) s += acc[k][r];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if (lane == 0) cyc[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = t1 - t0;
}

template <int MODE>
void run(const char* name, int threads) {
  const int blocks = 256, iters = 2000;
  float* out; long long* cyc;
  hipMalloc(&out, sizeof(float) * blocks * threads); hipMalloc(&cyc, sizeof(long long) * blocks * (threads / 64));
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, 10);
  hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(threads), 0, 0, out, cyc, iters);
  std::vector<long long> h(blocks * (threads / 64));
  hipMemcpy(h.data(), cyc, sizeof(long long) * h.size(), hipMemcpyDeviceToHost);
  double s = 0; for (long long v : h) s += (double)v;
  printf("%-44s %d waves/SIMD: %.1f cycles per MFMA per wave\n", name, threads / 256, s / h.size() / (iters * 24.0));
  hipFree(out); hipFree(cyc);
}

int main() {
  for (int threads : {256, 512}) {
    run<0>("chains of 6 into one accumulator", threads);
    run<1>("two accumulators alternating", threads);
    run<2>("four accumulators rotating", threads);
  }
  return 0;
}
