#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
__global__ void k(const float* x, float* o) {
  int i = threadIdx.x;
  float c0 = x[i], c1 = x[i], c2 = x[i], c3 = x[i];
  unsigned a = 0x40004080u;           // (lo = 4.0, hi = 2.0)
  unsigned s0 = 0x0000bf80u, s1 = 0xbf800000u;
  asm volatile("s_nop 4\n v_dot2c_f32_bf16 %0, %1, %2\n s_nop 4" : "+v"(c0) : "v"(a), "v"(s0));
  asm volatile("s_nop 4\n v_dot2c_f32_bf16 %0, %1, %2\n s_nop 4" : "+v"(c1) : "v"(a), "v"(s1));
  asm volatile("s_nop 4\n v_dot2c_f32_bf16 %0, %1, %2\n s_nop 4" : "+v"(c2) : "v"(s0), "v"(a));
  unsigned one = 0x3f803f80u;
  asm volatile("s_nop 4\n v_dot2c_f32_bf16 %0, %1, %2\n s_nop 4" : "+v"(c3) : "v"(a), "v"(one));
  o[4*i] = c0; o[4*i+1] = c1; o[4*i+2] = c2; o[4*i+3] = c3;
}
int main() {
  float hx[4] = {10.f, 100.f, 0.015625f, -3.f}, ho[16];
  float *dx, *dout; hipMalloc(&dx, 16); hipMalloc(&dout, 64);
  hipMemcpy(dx, hx, 16, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(4), 0, 0, dx, dout);
  hipMemcpy(ho, dout, 64, hipMemcpyDeviceToHost);
  for (int i = 0; i < 4; ++i) printf("c=%g: lo*-1 -> %g   hi*-1 -> %g   swapped -> %g   (1,1) -> %g\n", hx[i], ho[4*i], ho[4*i+1], ho[4*i+2], ho[4*i+3]);
}
