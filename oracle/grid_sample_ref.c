/* CPU ORACLE (C restatement) -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Bit-exact integer/index pieces of the FovealSeg path, restated in plain C so that every rounding is
 * explicit (fmaf, no contraction: build with -ffp-contract=off):
 *   - F.grid_sample(bilinear, zeros, align_corners=False) forward, ATen CPU arithmetic
 *     (reference call sites models/models.py:880,909; recipe SURVEY.md 8(a)-A12);
 *   - y_sampled.long() label map (models/models.py:951);
 *   - the truncated inverse index maps u,v (models/models.py:644-645).
 * Pinned by tests/test_oracle_c.py against tests/golden/g5_*.npz / g6_*.npz (outputs of the reference).
 */
#include <math.h>
#include <stdint.h>

static float sample_plane(const float* p, int H, int W, float gx, float gy) {
  const float ix = fmaf(gx + 1.0f, (float)W * 0.5f, -0.5f);
  const float iy = fmaf(gy + 1.0f, (float)H * 0.5f, -0.5f);
  const float fx = floorf(ix), fy = floorf(iy);
  const float w = ix - fx, e = 1.0f - w;
  const float n = iy - fy, s = 1.0f - n;
  const float nw = s * e, ne = s * w, sw = n * e, se = n * w;
  const float cx = fminf(fmaxf(fx, -2.0f), (float)W + 1.0f), cy = fminf(fmaxf(fy, -2.0f), (float)H + 1.0f);
  const int x0 = (int)cx, y0 = (int)cy;
  const int okx0 = x0 >= 0 && x0 < W, okx1 = x0 + 1 >= 0 && x0 + 1 < W;
  const int oky0 = y0 >= 0 && y0 < H, oky1 = y0 + 1 >= 0 && y0 + 1 < H;
  const float vnw = (oky0 && okx0) ? p[(long)y0 * W + x0] : 0.0f;
  const float vne = (oky0 && okx1) ? p[(long)y0 * W + x0 + 1] : 0.0f;
  const float vsw = (oky1 && okx0) ? p[(long)(y0 + 1) * W + x0] : 0.0f;
  const float vse = (oky1 && okx1) ? p[(long)(y0 + 1) * W + x0 + 1] : 0.0f;
  float acc = vnw * nw;
  acc = fmaf(vne, ne, acc);
  acc = fmaf(vsw, sw, acc);
  acc = fmaf(vse, se, acc);
  return acc;
}

/* x (B,C,H,W), grid (B,h,w,2) -> out (B,C,h,w) */
void fs_oracle_grid_sample(const float* x, const float* grid, float* out, int B, int C, int H, int W, int h, int w) {
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (long i = 0; i < (long)h * w; ++i) {
        const float* g = grid + ((long)b * h * w + i) * 2;
        out[((long)b * C + c) * h * w + i] = sample_plane(x + ((long)b * C + c) * H * W, H, W, g[0], g[1]);
      }
}

/* y (B,1,H,W) -> label (B,h,w) = (int64) bilinear sample, truncation toward zero */
void fs_oracle_label_map(const float* y, const float* grid, int64_t* label, int B, int H, int W, int h, int w) {
  for (int b = 0; b < B; ++b)
    for (long i = 0; i < (long)h * w; ++i) {
      const float* g = grid + ((long)b * h * w + i) * 2;
      label[(long)b * h * w + i] = (int64_t)sample_plane(y + (long)b * H * W, H, W, g[0], g[1]);
    }
}

/* u = int(((gx+1)/2)*(W-1)), v = int(((gy+1)/2)*(H-1)) */
void fs_oracle_inverse_index(const float* grid, int64_t* u, int64_t* v, long n, int H, int W) {
  for (long i = 0; i < n; ++i) {
    const float fu = ((grid[2 * i] + 1.0f) * 0.5f) * (float)(W - 1);
    const float fv = ((grid[2 * i + 1] + 1.0f) * 0.5f) * (float)(H - 1);
    u[i] = (int64_t)(int32_t)fu;
    v[i] = (int64_t)(int32_t)fv;
  }
}
