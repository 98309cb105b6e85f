// Is  x = h + m + l  exact when the three bf16 planes come from v_cvt_pk_bf16_f32 (round to nearest even) and the remainders from
// v_dot2c_f32_bf16 (D += a.lo * b.lo + a.hi * b.hi with b = (-1, 0) or (0, -1): the subtraction x - plane without unpacking the plane)?
// 7 VALU per value pair instead of the 15 of the integer split (conv_split.h PrecX3::split4).  Standalone:
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/split_probe tools/probes/split_probe.hip && /tmp/split_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <vector>
typedef float f2 __attribute__((ext_vector_type(2)));
typedef __bf16 b2 __attribute__((ext_vector_type(2)));
#ifndef PRE_N
#define PRE_N 0
#endif
#ifndef POST_N
#define POST_N 0
#endif
#if PRE_N == 0
#define PRE ""
#elif PRE_N == 1
#define PRE "s_nop 0\n"
#else
#define PRE "s_nop 1\n"
#endif
#if POST_N == 0
#define POST ""
#elif POST_N == 1
#define POST "\ns_nop 0"
#else
#define POST "\ns_nop 1"
#endif
__device__ __forceinline__ float dot2c(unsigned a, unsigned b, float c) {
  asm volatile(PRE "v_dot2c_f32_bf16 %0, %1, %2" POST : "+v"(c) : "v"(a), "v"(b));
  return c;
}
__global__ void split_kernel(const float* x, unsigned* o, long npair) {
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= npair) return;
  f2 v = {x[2 * i], x[2 * i + 1]};
  const unsigned hu = __builtin_bit_cast(unsigned, __builtin_convertvector(v, b2));
  const unsigned sel0 = 0x0000bf80u, sel1 = 0xbf800000u;
  f2 r = {dot2c(hu, sel0, v[0]), dot2c(hu, sel1, v[1])};
  const unsigned mu = __builtin_bit_cast(unsigned, __builtin_convertvector(r, b2));
  f2 s = {dot2c(mu, sel0, r[0]), dot2c(mu, sel1, r[1])};
  const unsigned lu = __builtin_bit_cast(unsigned, __builtin_convertvector(s, b2));
  o[3 * i] = hu; o[3 * i + 1] = mu; o[3 * i + 2] = lu;
}
static float bf(unsigned short b) { unsigned u = (unsigned)b << 16; float f; memcpy(&f, &u, 4); return f; }
int main() {
  const long npair = 1 << 22, n = 2 * npair;
  std::vector<float> x(n);
  srand(7);
  for (long i = 0; i < n; ++i) {
    unsigned u = ((unsigned)rand() << 16) ^ (unsigned)rand();
    const int kind = i & 7;
    if (kind < 5) { const unsigned e = 100 + rand() % 56; u = (u & 0x807fffffu) | (e << 23); }            // normal range around 1
    else if (kind == 5) { const unsigned e = 1 + rand() % 40; u = (u & 0x807fffffu) | (e << 23); }         // tiny normals: remainders go denormal
    else if (kind == 6) { u &= 0xffff0000u | (rand() & 0xffff); const unsigned e = 120 + rand() % 16; u = (u & 0x807fffffu) | (e << 23); if (rand() & 1) u &= 0xffff8000u; }   // ties
    else { const unsigned e = 200 + rand() % 50; u = (u & 0x807fffffu) | (e << 23); }                      // large
    memcpy(&x[i], &u, 4);
  }
  float* dx; unsigned* dout;
  hipMalloc(&dx, n * 4); hipMalloc(&dout, npair * 12);
  hipMemcpy(dx, x.data(), n * 4, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(split_kernel, dim3((unsigned)((npair + 255) / 256)), dim3(256), 0, 0, dx, dout, npair);
  std::vector<unsigned> o(npair * 3);
  if (hipMemcpy(o.data(), dout, npair * 12, hipMemcpyDeviceToHost) != hipSuccess) { printf("copy failed\n"); return 1; }
  long inexact = 0, inexact_tiny = 0, bad_hi = 0, worst_i = -1; double worst = 0;
  for (long i = 0; i < npair; ++i)
    for (int e = 0; e < 2; ++e) {
      const float xv = x[2 * i + e];
      const unsigned short h = (unsigned short)(e ? o[3 * i] >> 16 : o[3 * i] & 0xffff), m = (unsigned short)(e ? o[3 * i + 1] >> 16 : o[3 * i + 1] & 0xffff),
                           l = (unsigned short)(e ? o[3 * i + 2] >> 16 : o[3 * i + 2] & 0xffff);
      const double sum = (double)bf(h) + (double)bf(m) + (double)bf(l);
      if (sum != (double)xv) {
        const bool tiny = std::fabs(xv) < 1e-25f;
        if (tiny) ++inexact_tiny; else ++inexact;
        const double rel = std::fabs(sum - xv) / std::fabs(xv);
        if (!tiny && rel > worst) { worst = rel; worst_i = 2 * i + e; }
      }
      unsigned ux; memcpy(&ux, &xv, 4);
      const unsigned rne = (ux + 0x7fffu + ((ux >> 16) & 1u)) >> 16;
      if ((unsigned short)rne != h) ++bad_hi;
    }
  printf("values %ld  inexact (|x| >= 1e-25) %ld  inexact tiny %ld  first plane != RNE %ld  worst rel %.3e", n, inexact, inexact_tiny, bad_hi, worst);
  if (worst_i >= 0) printf("  at x = %.9g", x[worst_i]);
  printf("\n");
  return 0;
}
