// Diagnostic instrumentation of the 4-wave F(2,3) kernel (conv_wino.hip).  The shipped library is built WITHOUT these macros:
// every WINO_DIAG_* / WINO_STAMP below is then empty and nothing of this file reaches the binary.
//   -DFS_WINO_TRACE  per-phase time stamps of wave 0 of every workgroup's first tile, printed per launch (tools/wino_trace.sh)
//   -DFS_WINO_CLOCK  kernel-long deltas of s_memtime / s_memrealtime per workgroup = the shader clock under load
//                    (tools/wino_clock.sh; MI355X_MICROARCH.md, DVFS give-back (6)); read back through fs_debug_wino_clock_ghz()
#pragma once
#include <hip/hip_runtime.h>
#include <cstdio>

#if defined(FS_WINO_TRACE) || defined(FS_WINO_CLOCK)

#ifdef FS_WINO_TRACE
#define WINO_DIAG_FIELD_TRACE long long* dbg;      /* [workgroup][32] phase time stamps of wave 0 */
#define WINO_STAMP(i) do { if (tid == 0 && first_tile) a.dbg[(long)blockIdx.x * 32 + (i)] = clock64(); } while (0)
#define WINO_DIAG_TRACE_BEGIN bool first_tile = true
#define WINO_DIAG_NEXT_TILE first_tile = false
#else
#define WINO_DIAG_FIELD_TRACE
#define WINO_STAMP(i) do { } while (0)
#define WINO_DIAG_TRACE_BEGIN do { } while (0)
#define WINO_DIAG_NEXT_TILE do { } while (0)
#endif

#ifdef FS_WINO_CLOCK
#define WINO_DIAG_FIELD_CLOCK long long* clk;      /* [workgroup][2] deltas of s_memtime and s_memrealtime (100 MHz) */
#define WINO_DIAG_CLOCK_BEGIN const long long ck0 = clock64(), rt0 = wall_clock64()
#define WINO_DIAG_KERNEL_END \
  do { if (tid == 0 && a.clk != nullptr) { a.clk[2 * blockIdx.x] = clock64() - ck0; a.clk[2 * blockIdx.x + 1] = wall_clock64() - rt0; } } while (0)
static long long* g_wino_clk = nullptr;
static int g_wino_clk_n = 0;
#else
#define WINO_DIAG_FIELD_CLOCK
#define WINO_DIAG_CLOCK_BEGIN do { } while (0)
#define WINO_DIAG_KERNEL_END do { } while (0)
#endif

#define WINO_DIAG_FIELDS WINO_DIAG_FIELD_TRACE WINO_DIAG_FIELD_CLOCK
#define WINO_DIAG_KERNEL_BEGIN WINO_DIAG_TRACE_BEGIN; WINO_DIAG_CLOCK_BEGIN

template <class Args>
static inline int wino_diag_before_launch(Args& a, unsigned grid, hipStream_t stream) {
#ifdef FS_WINO_CLOCK
  if (g_wino_clk == nullptr && hipMalloc(&g_wino_clk, sizeof(long long) * 2 * 4096) != hipSuccess) return 1;
  a.clk = g_wino_clk; g_wino_clk_n = (int)grid;
#endif
#ifdef FS_WINO_TRACE
  static long long* dbg = nullptr;
  if (dbg == nullptr && hipMalloc(&dbg, sizeof(long long) * 32 * 65536) != hipSuccess) return 1;
  a.dbg = dbg;
  if (hipMemsetAsync(dbg, 0, sizeof(long long) * 32 * grid, stream) != hipSuccess) return 1;
#endif
  (void)stream; (void)grid;
  return 0;
}

template <class Args>
static inline int wino_diag_after_launch(const Args& a, unsigned grid, long ntile, hipStream_t stream) {
#ifdef FS_WINO_TRACE
  static long long host[32 * 65536];
  const long nwg = grid;
  if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(host, a.dbg, sizeof(long long) * 32 * nwg, hipMemcpyDeviceToHost) != hipSuccess) return 1;
  double sum[32] = {0};
  for (long w_ = 0; w_ < nwg; ++w_)
    for (int i = 1; i < 25; ++i) {
      if (host[w_ * 32 + i] == 0) continue;
      int prev = i - 1;
      while (prev > 0 && host[w_ * 32 + prev] == 0) --prev;
      sum[i] += (double)(host[w_ * 32 + i] - host[w_ * 32 + prev]);
    }
  fprintf(stderr, "wino trace (first tile of each workgroup) B%d %dx%d %d->%d nchunk %d tiles %ld grid %ld:", a.B, a.H, a.W, a.Cs, a.Cd, a.nchunk, ntile, nwg);
  for (int i = 1; i < 25; ++i) if (sum[i] > 0) fprintf(stderr, " [%d]%.0f", i, sum[i] / nwg);
  fprintf(stderr, "\n");
#endif
  (void)a; (void)grid; (void)ntile; (void)stream;
  return 0;
}
#define WINO_DIAG_BEFORE_LAUNCH(a, grid, stream) do { if (wino_diag_before_launch(a, grid, stream)) return FS_ERR_ARG; } while (0)
#define WINO_DIAG_AFTER_LAUNCH(a, grid, ntile, stream) do { if (wino_diag_after_launch(a, grid, ntile, stream)) return FS_ERR_ARG; } while (0)

#ifdef FS_WINO_CLOCK
// median over the workgroups of the LAST 4-wave launch of delta s_memtime / delta s_memrealtime (GHz)
extern "C" double fs_debug_wino_clock_ghz() {
  if (g_wino_clk == nullptr || g_wino_clk_n <= 0) return 0.0;
  static long long host[2 * 4096];
  if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(host, g_wino_clk, sizeof(long long) * 2 * g_wino_clk_n, hipMemcpyDeviceToHost) != hipSuccess) return 0.0;
  double r[4096]; int n = 0;
  for (int i = 0; i < g_wino_clk_n; ++i) if (host[2 * i + 1] > 0) r[n++] = (double)host[2 * i] / (double)host[2 * i + 1] * 0.1;
  if (n == 0) return 0.0;
  for (int i = 1; i < n; ++i) { double v = r[i]; int j = i - 1; while (j >= 0 && r[j] > v) { r[j + 1] = r[j]; --j; } r[j + 1] = v; }
  return r[n / 2];
}
#endif

#else      // shipped build

#define WINO_DIAG_FIELDS
#define WINO_STAMP(i) do { } while (0)
#define WINO_DIAG_KERNEL_BEGIN do { } while (0)
#define WINO_DIAG_NEXT_TILE do { } while (0)
#define WINO_DIAG_KERNEL_END do { } while (0)
#define WINO_DIAG_BEFORE_LAUNCH(a, grid, stream) do { } while (0)
#define WINO_DIAG_AFTER_LAUNCH(a, grid, ntile, stream) do { } while (0)

#endif
