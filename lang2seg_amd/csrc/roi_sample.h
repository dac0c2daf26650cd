// RoIAlign sampling shared by the crop kernels (roi.hip) and the fused RoIAlign -> layer4.0 kernel (conv_igemm.hip): one definition, so
// that both produce bit-identical crops.
#pragma once
#include "common.h"

struct Samp { int x0, y0; float wx1, wy1; };
__device__ __forceinline__ Samp roi_sample(const float* roi, int H, int W, int P, int py, int px, float sscale, float TH, float TW) {
#pragma clang fp contract(off)   // (no fused multiply-adds here: every kernel that inlines this must round the sample position the same way)
  // NET:122-147: theta from roi/16 over the map size, (TH, TW) = (H, W); NET:151-182 (_crop_pool_layer_align): theta from the roi in
  // image pixels over the image size, (TH, TW) = im_info and sscale = 1; affine_grid + grid_sample, align_corners = True
  const float x1 = roi[1] * sscale, y1 = roi[2] * sscale, x2 = roi[3] * sscale, y2 = roi[4] * sscale;
  const float t00 = (x2 - x1) / (TW - 1.f), t02 = (x1 + x2 - TW + 1.f) / (TW - 1.f);
  const float t11 = (y2 - y1) / (TH - 1.f), t12 = (y1 + y2 - TH + 1.f) / (TH - 1.f);
  const float step = 2.f / (float)(P - 1);
  // torch.linspace(-1, 1, P): start-based for i < P/2, end-based otherwise
  const float bx = (px < P / 2) ? (-1.f + step * (float)px) : (1.f - step * (float)(P - 1 - px));
  const float by = (py < P / 2) ? (-1.f + step * (float)py) : (1.f - step * (float)(P - 1 - py));
  const float gx = t00 * bx + t02, gy = t11 * by + t12;
  const float ix = ((gx + 1.f) / 2.f) * (float)(W - 1), iy = ((gy + 1.f) / 2.f) * (float)(H - 1);
  Samp s;
  const float fx = floorf(ix), fy = floorf(iy);
  s.x0 = (int)fx; s.y0 = (int)fy; s.wx1 = ix - fx; s.wy1 = iy - fy;
  return s;
}
// the four bilinear taps of one sample on 8 consecutive bf16 channels (16 bytes), fp32 blend, one rounding.  Branch-free: a tap outside
// the map keeps a valid (clamped) offset and gets weight 0 - fma(0, x, v) leaves v as it is for finite x - so that callers can issue the
// loads of several chunks before the first blend (a load behind a per-tap branch is waited for on the spot).
struct RoiTaps { int o00, o01, o10, o11; float w00, w01, w10, w11; };
__device__ __forceinline__ RoiTaps roi_taps(const Samp& s, int H, int W, int C) {
#pragma clang fp contract(off)
  RoiTaps t;
  const bool vx0 = s.x0 >= 0 && s.x0 < W, vx1 = s.x0 + 1 >= 0 && s.x0 + 1 < W, vy0 = s.y0 >= 0 && s.y0 < H, vy1 = s.y0 + 1 >= 0 && s.y0 + 1 < H;
  t.w00 = (vy0 && vx0) ? (1.f - s.wx1) * (1.f - s.wy1) : 0.f; t.w01 = (vy0 && vx1) ? s.wx1 * (1.f - s.wy1) : 0.f;
  t.w10 = (vy1 && vx0) ? (1.f - s.wx1) * s.wy1 : 0.f; t.w11 = (vy1 && vx1) ? s.wx1 * s.wy1 : 0.f;
  const int xa = min(max(s.x0, 0), W - 1), xb = min(max(s.x0 + 1, 0), W - 1), ya = min(max(s.y0, 0), H - 1), yb = min(max(s.y0 + 1, 0), H - 1);
  t.o00 = (ya * W + xa) * C; t.o01 = (ya * W + xb) * C; t.o10 = (yb * W + xa) * C; t.o11 = (yb * W + xb) * C;   // (elements: H*W*C < 2^30)
  return t;
}
struct RoiQuad { uint4 q00, q01, q10, q11; };
__device__ __forceinline__ RoiQuad roi_load8(const bf16_t* __restrict__ feat, const RoiTaps& t, int c) {
  RoiQuad q;
  q.q00 = *(const uint4*)(feat + t.o00 + c); q.q01 = *(const uint4*)(feat + t.o01 + c);
  q.q10 = *(const uint4*)(feat + t.o10 + c); q.q11 = *(const uint4*)(feat + t.o11 + c);
  return q;
}
__device__ __forceinline__ uint4 roi_mix8(const RoiQuad& q, const RoiTaps& t) {
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] = 0.f;
  auto tap = [&](const uint4& x, float w) {
    const uint32_t qw[4] = {x.x, x.y, x.z, x.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = fmaf(w, __uint_as_float(qw[e] << 16), v[2 * e]); v[2 * e + 1] = fmaf(w, __uint_as_float(qw[e] & 0xFFFF0000u), v[2 * e + 1]); }
  };
  tap(q.q00, t.w00); tap(q.q01, t.w01); tap(q.q10, t.w10); tap(q.q11, t.w11);
  uint4 o;
  o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
  o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
  return o;
}
__device__ __forceinline__ uint4 roi_blend8(const bf16_t* __restrict__ feat, const RoiTaps& t, int c) { return roi_mix8(roi_load8(feat, t, c), t); }
