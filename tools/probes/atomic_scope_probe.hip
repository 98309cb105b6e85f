// Probe: cost of the split-K reduction of the bwd-weight kernel (512 workgroups x 36 864 fp32 adds into one 147 KB tile) done with
//   A  agent-scope atomics into ONE tile (what conv_wgrad_class_kernel does)
//   B  the same atomics into a tile per XCD (XCC_ID hardware register), summed afterwards
//   C  plain stores of the partial tiles (75 MB), summed afterwards
//   D  as A with 8x fewer workgroups' worth of atomics (lower bound of B)
// Build: hipcc -O3 --offload-arch=gfx950 tools/probes/atomic_scope_probe.hip -o gpurun_out/atomic_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int TILE = 36864;   // 9 taps x 64 x 64
__device__ __forceinline__ int xcc_id() { return (int)(__builtin_amdgcn_s_getreg((20 /*HW_REG_XCC_ID*/) | (0 << 6) | (3 << 11)) & 7); }
__global__ __launch_bounds__(256) void k_atomic(float* dw, int per_xcd, int* seen) {
  float* dst = dw + (per_xcd ? (long)xcc_id() * TILE : 0);
  if (threadIdx.x == 0 && seen) atomicOr(&seen[blockIdx.x & 7], 1 << xcc_id());
  for (int i = threadIdx.x; i < TILE; i += 256) atomicAdd(&dst[i], 1.0f);
}
__global__ __launch_bounds__(256) void k_store(float* slab) {
  float* dst = slab + (long)blockIdx.x * TILE;
  for (int i = threadIdx.x; i < TILE; i += 256) __builtin_nontemporal_store(1.0f, &dst[i]);
}
__global__ __launch_bounds__(256) void k_reduce(const float* slab, float* dw, int n) {
  const int i = blockIdx.x * 256 + threadIdx.x;
  if (i >= TILE) return;
  float s = 0.f;
  for (int k = 0; k < n; ++k) s += slab[(long)k * TILE + i];
  dw[i] += s;
}
int main() {
  const int NWG = 512;
  float *dw, *slab; int* seen;
  hipMalloc(&dw, 8L * TILE * 4); hipMalloc(&slab, (long)NWG * TILE * 4); hipMalloc(&seen, 32);
  hipMemset(dw, 0, 8L * TILE * 4); hipMemset(seen, 0, 32);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  auto time = [&](const char* name, auto fn) {
    fn(); hipDeviceSynchronize();
    hipEventRecord(e0); for (int r = 0; r < 20; ++r) fn(); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-44s %8.1f us\n", name, ms * 1000 / 20);
  };
  time("A one tile, 512 WGs of atomics", [&] { k_atomic<<<NWG, 256>>>(dw, 0, nullptr); });
  time("B tile per XCD (XCC_ID) + reduce of 8", [&] { k_atomic<<<NWG, 256>>>(dw, 1, seen); k_reduce<<<TILE / 256, 256>>>(dw + TILE, dw, 7); });
  time("B' tile per XCD, atomics only", [&] { k_atomic<<<NWG, 256>>>(dw, 1, nullptr); });
  time("C stores of 512 partial tiles + reduce", [&] { k_store<<<NWG, 256>>>(slab); k_reduce<<<TILE / 256, 256>>>(slab, dw, NWG); });
  time("C' stores only", [&] { k_store<<<NWG, 256>>>(slab); });
  time("D one tile, 64 WGs of atomics", [&] { k_atomic<<<64, 256>>>(dw, 0, nullptr); });
  // check B: every address of the 8 tiles must hold an integer count, and the counts over tiles must sum to the launches made
  hipMemset(dw, 0, 8L * TILE * 4); k_atomic<<<NWG, 256>>>(dw, 1, seen); hipDeviceSynchronize();
  std::vector<float> h(8 * TILE); hipMemcpy(h.data(), dw, 8L * TILE * 4, hipMemcpyDeviceToHost);
  int hs[8]; hipMemcpy(hs, seen, 32, hipMemcpyDeviceToHost);
  double tot = 0; bool flat = true;
  for (int x = 0; x < 8; ++x) { tot += h[x * TILE]; for (int i = 0; i < TILE; ++i) flat &= h[x * TILE + i] == h[x * TILE]; printf("xcd %d: %g workgroups; blockIdx&7==%d ran on XCC mask 0x%x\n", x, h[x * TILE], x, hs[x]); }
  printf("sum over XCD tiles %g (expect %d), tiles uniform: %d\n", tot, NWG, (int)flat);
  return 0;
}
