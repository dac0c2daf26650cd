// Input side of the train step: what lib/loaders/cycle_loader.py does per image on one CPU thread
// (cv2 resize of the mean-subtracted image, COCO run-length mask decode, union over segments, PIL nearest resize) as
// HBM-bound kernels fed with the raw bytes: the uint8 image and the run lengths are all that crosses PCIe.
//   l2s_rle_from_string   host     pyutils/refer/external/maskApi.c:217-231 (rleFrString)
//   l2s_prep_geometry     host     pyutils/mask-faster-rcnn/lib/utils/blob.py:35-45 + cv2.resize's dsize rule
//   l2s_prep_image        device   blob.py:32-47 (prep_im_for_blob: astype(float32) - means, cv2.resize INTER_LINEAR)
//   l2s_rle_to_mask       device   maskApi.c:43-47 (rleDecode) + cycle_loader.py:205-209 (sum over segments > 0, imresize nearest)
#include "common.h"
#include "../../include/lang2seg_hip.h"
#include <cmath>

namespace {

// ---------------------------------------------------------------- run-length masks
// inclusive prefix sums of the run lengths of each RLE object (one workgroup per object, 1024-wide chunks with a carry)
__global__ __launch_bounds__(1024) void rle_scan_kernel(const uint32_t* cnts, const int* offs, uint32_t* pre) {
  __shared__ uint32_t sh[1024];
  __shared__ uint32_t carry;
  const int o0 = offs[blockIdx.x], m = offs[blockIdx.x + 1] - o0, t = threadIdx.x;
  if (t == 0) carry = 0;
  __syncthreads();
  for (int base = 0; base < m; base += 1024) {
    const int i = base + t;
    sh[t] = i < m ? cnts[o0 + i] : 0u;
    __syncthreads();
    for (int o = 1; o < 1024; o <<= 1) {
      const uint32_t v = t >= o ? sh[t - o] : 0u;
      __syncthreads();
      sh[t] += v;
      __syncthreads();
    }
    const uint32_t c = carry;
    if (i < m) pre[o0 + i] = sh[t] + c;
    __syncthreads();
    if (t == 1023) carry = c + sh[1023];
    __syncthreads();
  }
}
// PIL NEAREST source index of every output row / column (oracle/boxes.py nearest_index: xo = 0.5 s, then += s in float64)
__global__ void nearest_table_kernel(int h, int oh, int w, int ow, int* ty, int* tx) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k < oh) {
    const double s = (double)h / (double)oh;
    double xo = 0.5 * s;
    for (int i = 0; i < k; ++i) xo += s;
    ty[k] = min((int)xo, h - 1);
  }
  if (k < ow) {
    const double s = (double)w / (double)ow;
    double xo = 0.5 * s;
    for (int i = 0; i < k; ++i) xo += s;
    tx[k] = min((int)xo, w - 1);
  }
}
// out[y][x] = 1 if any object covers source pixel (ty[y], tx[x]); an object's runs are over the column-major index x*h + y and
// alternate 0,1,0,... (maskApi.c:43-47), so the pixel's value is the parity of the run that holds its index (binary search)
__global__ __launch_bounds__(256) void rle_mask_kernel(const uint32_t* pre, const int* offs, int n, int h, const int* ty, const int* tx,
                                                       int oh, int ow, uint8_t* out) {
  const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
  if (x >= ow) return;
  const uint32_t idx = (uint32_t)tx[x] * (uint32_t)h + (uint32_t)ty[y];
  int v = 0;
  for (int r = 0; r < n && !v; ++r) {
    const int o0 = offs[r], m = offs[r + 1] - o0;
    int lo = 0, hi = m;                       // first j with pre[j] > idx
    while (lo < hi) {
      const int mid = (lo + hi) >> 1;
      if (pre[o0 + mid] > idx) hi = mid; else lo = mid + 1;
    }
    v = (lo < m) & (lo & 1);
  }
  out[(long)y * ow + x] = (uint8_t)v;
}

// ---------------------------------------------------------------- image
// cv2.resize(float32 image, fx = fy = scale, INTER_LINEAR): horizontal pass S[sx] a0 + S[sx+1] a1 with
// fx = float((dx + 0.5) / scale - 0.5), sx = floor(fx), a1 = fx - sx (zeroed, sx clamped at both borders), then the
// vertical pass R0 b0 + R1 b1 on rows clamped into the image; every product and sum rounded to float32 separately.
__device__ __forceinline__ void lin_coef(int d, double inv_scale, int n_src, bool clamp_w, int& s0, int& s1, float& a0, float& a1) {
  float f = (float)(((double)d + 0.5) * inv_scale - 0.5);
  int s = (int)floorf(f);
  f -= (float)s;
  if (clamp_w) {                               // columns: weights are edited at the borders
    if (s < 0) { f = 0.f; s = 0; }
    if (s >= n_src - 1) { f = 0.f; s = n_src - 1; }
    s0 = s; s1 = min(s + 1, n_src - 1);
  } else {                                     // rows: the two row indices are clipped, the weights are kept
    s0 = min(max(s, 0), n_src - 1); s1 = min(max(s + 1, 0), n_src - 1);
  }
  a0 = 1.f - f; a1 = f;
}
__global__ __launch_bounds__(256) void prep_image_kernel(const uint8_t* img, int h, int w, double m0, double m1, double m2, double inv_scale,
                                                         int oh, int ow, float* out) {
  const int ox = blockIdx.x * blockDim.x + threadIdx.x, oy = blockIdx.y;
  if (ox >= ow) return;
  int x0, x1, y0, y1; float a0, a1, b0, b1;
  lin_coef(ox, inv_scale, w, true, x0, x1, a0, a1);
  lin_coef(oy, inv_scale, h, false, y0, y1, b0, b1);
  const double mean[3] = {m0, m1, m2};
  const uint8_t* r0 = img + (long)y0 * w * 3; const uint8_t* r1 = img + (long)y1 * w * 3;
  float* o = out + ((long)oy * ow + ox) * 3;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
#pragma clang fp contract(off)                 // cv2's two passes round every product and sum: no fused multiply-add here
    const float p00 = (float)((double)r0[x0 * 3 + c] - mean[c]), p01 = (float)((double)r0[x1 * 3 + c] - mean[c]);
    const float p10 = (float)((double)r1[x0 * 3 + c] - mean[c]), p11 = (float)((double)r1[x1 * 3 + c] - mean[c]);
    const float h0 = p00 * a0 + p01 * a1;
    const float h1 = p10 * a0 + p11 * a1;
    o[c] = h0 * b0 + h1 * b1;
  }
}

}  // namespace

extern "C" int l2s_rle_from_string(const char* s, uint32_t* cnts, int max_counts) {
  if (!s || !cnts) return -1;
  long m = 0; size_t p = 0;
  while (s[p]) {
    long x = 0; int k = 0, more = 1;
    while (more) {
      if (!s[p]) return -1;                       // truncated string
      const char c = (char)(s[p] - 48);
      x |= (long)(c & 0x1f) << (5 * k);
      more = c & 0x20; ++p; ++k;
      if (!more && (c & 0x10)) x |= -1L << (5 * k);
    }
    if (m > 2) x += (long)cnts[m - 2];
    if (m >= max_counts) return -1;
    cnts[m++] = (uint32_t)x;
  }
  return (int)m;
}

extern "C" int l2s_prep_geometry(int h, int w, int target_size, int max_size, double* scale, int* oh, int* ow) {
  if (h <= 0 || w <= 0 || !scale || !oh || !ow) return L2S_EINVAL;
  const int smin = h < w ? h : w, smax = h < w ? w : h;
  double sc = (double)target_size / (double)smin;
  if (std::nearbyint(sc * (double)smax) > (double)max_size) sc = (double)max_size / (double)smax;     // np.round: half to even
  *scale = sc;
  *oh = (int)std::nearbyint((double)h * sc);                                                         // cvRound
  *ow = (int)std::nearbyint((double)w * sc);
  return L2S_OK;
}

extern "C" int l2s_prep_image(const uint8_t* img_bgr, int h, int w, double mean_b, double mean_g, double mean_r, double scale,
                              int oh, int ow, float* out, hipStream_t s) {
  if (h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || !(scale > 0.0)) return L2S_EINVAL;
  L2S_LAUNCH(prep_image_kernel, dim3(cdiv(ow, 256), oh), dim3(256), 0, s, img_bgr, h, w, mean_b, mean_g, mean_r, 1.0 / scale, oh, ow, out);
  return l2s_check_launch();
}

extern "C" long l2s_rle_ws_words(int total_counts, int oh, int ow) { return (long)total_counts + oh + ow + 16; }

extern "C" int l2s_rle_to_mask(const uint32_t* cnts, const int* offs, int n, int total_counts, int h, int w, int oh, int ow,
                               uint32_t* ws, uint8_t* out, hipStream_t s) {
  if (n < 1 || h <= 0 || w <= 0 || oh <= 0 || ow <= 0 || (long)h * w >= (1L << 32)) return L2S_EINVAL;
  uint32_t* pre = ws;
  int* ty = (int*)(ws + total_counts); int* tx = ty + oh;
  L2S_LAUNCH(rle_scan_kernel, dim3(n), dim3(1024), 0, s, cnts, offs, pre);
  L2S_LAUNCH(nearest_table_kernel, dim3(cdiv(oh > ow ? oh : ow, 256)), dim3(256), 0, s, h, oh, w, ow, ty, tx);
  L2S_LAUNCH(rle_mask_kernel, dim3(cdiv(ow, 256), oh), dim3(256), 0, s, (const uint32_t*)pre, offs, n, h, (const int*)ty, (const int*)tx, oh, ow, out);
  return l2s_check_launch();
}
