// Detection / mask losses of the lang2seg train step with fused gradients (gfx950).
// Reference: pyutils/mask-faster-rcnn/lib/nets/network_cycle_res5_2.py:360-413,448.
// Small reductions: latency-bound; each kernel writes its loss into the shared float loss[8] buffer
// and the gradient w.r.t. the head outputs in the activation dtype (consumed by the igemm dgrad/wgrad).
#include "common.h"
#include "../../include/lang2seg_hip.h"

namespace {

__device__ __forceinline__ float smooth_l1(float d, float s2, float& grad) {
  const float ad = fabsf(d);
  if (ad < 1.f / s2) { grad = s2 * d; return 0.5f * s2 * d * d; }
  grad = d > 0.f ? 1.f : (d < 0.f ? -1.f : 0.f);
  return ad - 0.5f / s2;
}

// multi-workgroup: the number of sampled anchors comes from the anchor-target kernel (device int), so no counting pass
__global__ __launch_bounds__(256) void rpn_loss_kernel(const float* heads, int ldh, const int* labels, const float* tgt, const float* inw,
                                                      const float* outw, int H, int W, int A, float sigma, float gscale, float* loss,
                                                      void* dheads, int ldd, int dt, const int* count_dev) {
  __shared__ float red[4];
  const int n = H * W * A;
  const float cnt = (float)(*count_dev);
  const float inv = cnt > 0.f ? 1.f / cnt : 0.f;
  const float s2 = sigma * sigma;
  float lce = 0.f, lbox = 0.f;
  for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    // i enumerates (h, w, a); the label array is laid out (a, h, w)
    const int a = i % A, pix = i / A, w = pix % W, h = pix / W;
    const int l = labels[(a * H + h) * W + w];
    const float* hr = heads + (long)pix * ldh;
    float dbg = 0.f, dfg = 0.f;
    if (l != -1) {
      const float bg = hr[a], fg = hr[A + a];
      const float m = fmaxf(bg, fg);
      const float lse = m + logf(expf(bg - m) + expf(fg - m));
      lce += lse - (l == 1 ? fg : bg);
      const float pb = expf(bg - lse), pf = expf(fg - lse);
      dbg = (pb - (l == 0 ? 1.f : 0.f)) * inv;
      dfg = (pf - (l == 1 ? 1.f : 0.f)) * inv;
    }
    stx(dheads, (long)pix * ldd + a, dt, dbg * gscale);
    stx(dheads, (long)pix * ldd + A + a, dt, dfg * gscale);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long ti = (long)i * 4 + k;
      const float iw = inw[ti], ow = outw[ti];
      float g;
      const float v = smooth_l1(iw * (hr[2 * A + a * 4 + k] - tgt[ti]), s2, g);
      lbox += ow * v;
      stx(dheads, (long)pix * ldd + 2 * A + a * 4 + k, dt, ow * iw * g * gscale);
    }
    // padding columns of dheads (so the dgrad GEMM can run over ldd columns): written by the a == 0 thread of the pixel
    if (a == 0) for (int c = 6 * A; c < ldd; ++c) stx(dheads, (long)pix * ldd + c, dt, 0.f);
  }
  lce = block_sum(lce, red);
  lbox = block_sum(lbox, red);
  if (threadIdx.x == 0) { atomicAdd(loss + L2S_LOSS_RPN_CLS, lce * inv); atomicAdd(loss + L2S_LOSS_RPN_BOX, lbox); }
}
__global__ void count_labels_kernel(const int* labels, int n, int* out) {   // fallback when no device count is supplied
  __shared__ float red[16];
  float c = 0.f;
  for (int i = threadIdx.x; i < n; i += blockDim.x) c += labels[i] != -1 ? 1.f : 0.f;
  c = block_sum(c, red);
  if (threadIdx.x == 0) *out = (int)c;
}

// one wave per roi
__global__ __launch_bounds__(256) void rcnn_loss_kernel(const float* heads, int ldh, const int* labels, const float* bt, const float* bi,
                                                       const float* bo, int R, int ncls, float gscale, float* loss, void* dheads, int ldd, int dt) {
  // (round 5: the two loss terms of a workgroup's four RoIs are summed in LDS and added with ONE atomic each - 512 float atomics on two
  // addresses serialised in the L2 for ~10 of the launch's 15 us)
  __shared__ float part[4][2];
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  if (lane == 0) { part[wv][0] = 0.f; part[wv][1] = 0.f; }
  if (r < R) {
  const float* hr = heads + (long)r * ldh;
  const int lab = labels[r];
  float mx = -INFINITY;
  for (int c = lane; c < ncls; c += 64) mx = fmaxf(mx, hr[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < ncls; c += 64) se += expf(hr[c] - mx);
  se = wave_sum(se);
  const float lse = mx + logf(se);
  const float invR = 1.f / (float)R;
  for (int c = lane; c < ncls; c += 64)
    stx(dheads, (long)r * ldd + c, dt, (expf(hr[c] - lse) - (c == lab ? 1.f : 0.f)) * invR * gscale);
  float lb = 0.f;
  for (int c = lane; c < 4 * ncls; c += 64) {
    const long ti = (long)r * 4 * ncls + c;
    float g;
    const float v = smooth_l1(bi[ti] * (hr[ncls + c] - bt[ti]), 1.f, g);
    lb += bo[ti] * v;
    stx(dheads, (long)r * ldd + ncls + c, dt, bo[ti] * bi[ti] * g * invR * gscale);
  }
  for (int c = 5 * ncls + lane; c < ldd; c += 64) stx(dheads, (long)r * ldd + c, dt, 0.f);
  lb = wave_sum(lb);
  if (lane == 0) { part[wv][0] = (lse - hr[lab]) * invR; part[wv][1] = lb * invR; }
  }
  __syncthreads();
  if (threadIdx.x < 2) atomicAdd(loss + (threadIdx.x ? L2S_LOSS_BOX : L2S_LOSS_CLS), (part[0][threadIdx.x] + part[1][threadIdx.x]) + (part[2][threadIdx.x] + part[3][threadIdx.x]));
}

__global__ __launch_bounds__(256) void mask_loss_kernel(const float* score, int ldsc, const int* labels, const float* mt, const int* num_fg,
                                                       int fg_max, int ms2, float gscale, float* loss, float* dscore) {
  __shared__ float red[4];
  const int nfg = min(*num_fg, fg_max);
  const float inv = nfg > 0 ? 1.f / (float)(nfg * ms2) : 0.f;
  float l = 0.f;
  for (int e = blockIdx.x * blockDim.x + threadIdx.x; e < fg_max * ms2; e += gridDim.x * blockDim.x) {
    const int s = e / ms2;
    float d = 0.f;
    if (s < nfg) {
      const float x = score[(long)e * ldsc + labels[s]], t = mt[e];
      l += fmaxf(x, 0.f) - x * t + log1pf(expf(-fabsf(x)));          // BCE with logits (NET:413)
      d = (1.f / (1.f + expf(-x)) - t) * inv * gscale;
    }
    dscore[e] = d;
  }
  l = block_sum(l, red);
  if (threadIdx.x == 0) atomicAdd(loss + L2S_LOSS_MASK, l * inv);
}

__global__ void total_loss_kernel(float* loss, float cap_w) {
  loss[L2S_LOSS_TOTAL] = loss[L2S_LOSS_CLS] + loss[L2S_LOSS_BOX] + loss[L2S_LOSS_RPN_CLS] + loss[L2S_LOSS_RPN_BOX] +
                         loss[L2S_LOSS_MASK] + loss[L2S_LOSS_RESPONSE] + cap_w * loss[L2S_LOSS_CAP];
}

// response loss (network_cycle_response.py:415-423): BCE-with-logits between the raw response map [H][W] and the GT mask
// resized to it by scipy.misc.imresize(..., interp='nearest') = PIL NEAREST: source index (int)xo with xo = 0.5 s, += s in
// float64 (sequential adds, as in the mask targets of proposal_target_layer.py:196)
__device__ __forceinline__ int pil_nearest_index(int n_in, int n_out, int k) {
  const double s = (double)n_in / (double)n_out;
  double xo = 0.5 * s;
  for (int i = 0; i < k; ++i) xo += s;
  return min((int)xo, n_in - 1);
}
__global__ __launch_bounds__(256) void response_loss_kernel(const float* resp, const uint8_t* mask, int MH, int MW, int H, int W, float gscale,
                                                           float* loss, float* dresp) {
  __shared__ float red[4];
  const int n = H * W;
  const float inv = 1.f / (float)n;
  float l = 0.f;
  for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < n; p += gridDim.x * blockDim.x) {
    const int y = p / W, x = p - y * W;
    const float t = (float)mask[(long)pil_nearest_index(MH, H, y) * MW + pil_nearest_index(MW, W, x)];
    const float v = resp[p];
    l += fmaxf(v, 0.f) - v * t + log1pf(expf(-fabsf(v)));
    dresp[p] = (1.f / (1.f + expf(-v)) - t) * inv * gscale;
  }
  l = block_sum(l, red);
  if (threadIdx.x == 0) atomicAdd(loss + L2S_LOSS_RESPONSE, l * inv);
}

// grid (roi, pixel chunk): dx[p][c] = dscore[p] * W[label][c] (ReLU-masked by x); part[roi][chunk][c] = sum_{p in chunk} dscore[p] x[p][c],
// part[roi][chunk][C] = sum_p dscore[p]; the second kernel adds the partial sums into dW[label] / db[label] in (RoI, chunk) order (all
// foreground RoIs of a step usually share one label: a single owner per (label, channel), no atomics, bit-reproducible)
constexpr int MPB_CHUNKS = 14;
__global__ __launch_bounds__(256) void maskpred_bwd_kernel(const float* dscore, const int* labels, const int* num_fg, int fg_max, int ms2, int C,
                                                          const float* w, const void* x, const void* ref, void* dx, float* part, int dt) {
  const int s = blockIdx.x;
  const int pc = (ms2 + gridDim.y - 1) / gridDim.y, p0 = blockIdx.y * pc, p1 = min(ms2, p0 + pc);
  const int nfg = min(*num_fg, fg_max);
  const bool valid = s < nfg;
  const int lab = valid ? labels[s] : 0;
  float* po = part + ((long)s * gridDim.y + blockIdx.y) * (C + 1);
  for (int c = threadIdx.x; c < C; c += blockDim.x) {
    const float wv = w[(long)lab * C + c];
    float acc = 0.f;
    for (int p = p0; p < p1; ++p) {
      const long o = ((long)s * ms2 + p) * C + c;
      const float d = valid ? dscore[s * ms2 + p] : 0.f;
      const float xv = ldx(x, o, dt);
      acc = fmaf(d, xv, acc);
      float g = d * wv;
      if (ref && !(ldx(ref, o, dt) > 0.f)) g = 0.f;
      stx(dx, o, dt, g);
    }
    po[c] = acc;
  }
  if (threadIdx.x == 0) {
    float sb = 0.f;
    if (valid) for (int p = p0; p < p1; ++p) sb += dscore[s * ms2 + p];
    po[C] = sb;
  }
}
// The per-(RoI, pixel chunk) partial sums of maskpred_bwd_kernel, added per label in RoI order (no atomics: two RoIs of one label add in a fixed
// order).  One thread per column c (the C weight columns + the bias).  Round 6: the running sums live in LDS ([ncls_max][256] floats per
// workgroup, thread c touches column c only: conflict-free) instead of in dW itself - the old form read, added to and stored dW[label][c] once per
// RoI, a store -> load round trip through memory per RoI (27 us for 64 RoIs); dW / db are added to ONCE per label at the end.  The sums are
// formed in the same order from the same zero, so with dW cleared beforehand (it is: this kernel is its only writer) the bits are the same.
constexpr int MPB_NCLS = 96;                              // label rows of the LDS table (mask_pred_net has 81 outputs)
__global__ __launch_bounds__(256) void maskpred_bwd_reduce_kernel(const float* part, const int* labels, const int* num_fg, int fg_max, int C, int chunks,
                                                                 float* dw, float* db) {
  __shared__ float acc[MPB_NCLS][256];
  __shared__ int used[MPB_NCLS];
  const int nfg = min(*num_fg, fg_max), t = threadIdx.x;
  const int c = blockIdx.x * blockDim.x + t;
  for (int l = 0; l < MPB_NCLS; ++l) acc[l][t] = 0.f;
  if (t < MPB_NCLS) used[t] = 0;
  __syncthreads();
  if (c <= C)
    for (int s = 0; s < nfg; ++s) {
      float v = 0.f;
      for (int q = 0; q < chunks; ++q) v += part[((long)s * chunks + q) * (C + 1) + c];
      const int l = labels[s];
      if (l >= 0 && l < MPB_NCLS) { acc[l][t] += v; if (t == 0) used[l] = 1; }
      else if (c < C) dw[(long)l * C + c] += v; else db[l] += v;          // (a label beyond the table: the old path)
    }
  __syncthreads();
  if (c <= C)
    for (int l = 0; l < MPB_NCLS; ++l)
      if (used[l]) { if (c < C) dw[(long)l * C + c] += acc[l][t]; else db[l] += acc[l][t]; }
}

// ---- TEST-mode heads (NET:277-307, 650-658): class probabilities, de-normalised box deltas, mask probabilities ----
// one wave per roi
__global__ __launch_bounds__(256) void rcnn_predict_kernel(const float* heads, int ldh, int R, int ncls, const float* stds, const float* means,
                                                          float* cls_prob, float* bbox_pred) {
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (r >= R) return;
  const float* hr = heads + (long)r * ldh;
  float mx = -INFINITY;
  for (int c = lane; c < ncls; c += 64) mx = fmaxf(mx, hr[c]);
  mx = wave_max(mx);
  float se = 0.f;
  for (int c = lane; c < ncls; c += 64) se += expf(hr[c] - mx);
  se = wave_sum(se);
  for (int c = lane; c < ncls; c += 64) cls_prob[(long)r * ncls + c] = expf(hr[c] - mx) / se;
  for (int c = lane; c < 4 * ncls; c += 64) bbox_pred[(long)r * 4 * ncls + c] = hr[ncls + c] * stds[c & 3] + means[c & 3];
}
// labels == nullptr: out[e][c] = sigmoid(score[e][c]) for every class; else out[e] = sigmoid(score[e][labels[e / ms2]])
__global__ void mask_prob_kernel(const float* score, int ldsc, int ncls, const int* labels, int ms2, long n_elem, float* out) {
  const long total = labels ? n_elem : n_elem * ncls;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v;
    if (labels) v = score[i * ldsc + labels[i / ms2]];
    else { const long e = i / ncls; v = score[e * ldsc + (i - e * ncls)]; }
    out[i] = 1.f / (1.f + expf(-v));
  }
}

}  // namespace

extern "C" int l2s_rpn_loss(const float* heads, int ldh, const int* labels, const float* targets, const float* inside_w,
                            const float* outside_w, int H, int W, int A, float sigma, float gscale, float* loss, void* dheads, int ldd,
                            int dtype, const int* count_dev, int* count_ws, hipStream_t s) {
  const int n = H * W * A;
  if (!count_dev) {
    if (!count_ws) return L2S_EINVAL;
    L2S_LAUNCH(count_labels_kernel, dim3(1), dim3(1024), 0, s, labels, n, count_ws);
    count_dev = count_ws;
  }
  L2S_LAUNCH(rpn_loss_kernel, dim3(cdiv(n, 256)), dim3(256), 0, s, heads, ldh, labels, targets, inside_w, outside_w, H, W, A, sigma, gscale, loss, dheads, ldd, dtype, count_dev);
  return l2s_check_launch();
}
extern "C" int l2s_rcnn_loss(const float* heads, int ldh, const int* labels, const float* bbox_targets, const float* inside_w,
                             const float* outside_w, int R, int ncls, float gscale, float* loss, void* dheads, int ldd, int dtype, hipStream_t s) {
  L2S_LAUNCH(rcnn_loss_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, heads, ldh, labels, bbox_targets, inside_w, outside_w, R, ncls, gscale, loss, dheads, ldd, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_mask_loss(const float* score, int ldsc, const int* labels, const float* mask_targets, const int* num_fg, int fg_max,
                             int ms2, float gscale, float* loss, float* dscore, hipStream_t s) {
  L2S_LAUNCH(mask_loss_kernel, dim3(cdiv(fg_max * ms2, 256)), dim3(256), 0, s, score, ldsc, labels, mask_targets, num_fg, fg_max, ms2, gscale, loss, dscore);
  return l2s_check_launch();
}
extern "C" int l2s_response_loss(const float* resp, const uint8_t* gt_mask, int mask_h, int mask_w, int H, int W, float gscale, float* loss,
                                 float* dresp, hipStream_t s) {
  L2S_LAUNCH(response_loss_kernel, dim3(cdiv(H * W, 256)), dim3(256), 0, s, resp, gt_mask, mask_h, mask_w, H, W, gscale, loss, dresp);
  return l2s_check_launch();
}
extern "C" int l2s_total_loss(float* loss, float cap_w, hipStream_t s) {
  L2S_LAUNCH(total_loss_kernel, dim3(1), dim3(1), 0, s, loss, cap_w);
  return l2s_check_launch();
}
extern "C" long l2s_maskpred_ws_floats(int fg_max, int C) { return (long)fg_max * MPB_CHUNKS * (C + 1); }
extern "C" int l2s_maskpred_bwd(const float* dscore, const int* labels, const int* num_fg, int fg_max, int ms2, int C, const float* w,
                                const void* x, const void* relu_ref, void* dx, float* dw, float* db, float* ws, int dtype, hipStream_t s) {
  if (!ws) return L2S_EINVAL;                            // l2s_maskpred_ws_floats(fg_max, C) floats
  L2S_LAUNCH(maskpred_bwd_kernel, dim3(fg_max, MPB_CHUNKS), dim3(256), 0, s, dscore, labels, num_fg, fg_max, ms2, C, w, x, relu_ref, dx, ws, dtype);
  L2S_LAUNCH(maskpred_bwd_reduce_kernel, dim3(cdiv(C + 1, 256)), dim3(256), 0, s, (const float*)ws, labels, num_fg, fg_max, C, MPB_CHUNKS, dw, db);
  return l2s_check_launch();
}
// the two launches of l2s_maskpred_bwd separately: the data gradient (+ the partial sums into ws) on the caller's critical path, the ordered reduction of
// the partial sums into dW / db wherever the caller has room (the weight-gradient stream: nothing of the backward chain reads dW)
extern "C" int l2s_maskpred_bwd_dx(const float* dscore, const int* labels, const int* num_fg, int fg_max, int ms2, int C, const float* w,
                                   const void* x, const void* relu_ref, void* dx, float* ws, int dtype, hipStream_t s) {
  if (!ws) return L2S_EINVAL;
  L2S_LAUNCH(maskpred_bwd_kernel, dim3(fg_max, MPB_CHUNKS), dim3(256), 0, s, dscore, labels, num_fg, fg_max, ms2, C, w, x, relu_ref, dx, ws, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_maskpred_bwd_reduce(const float* ws, const int* labels, const int* num_fg, int fg_max, int C, float* dw, float* db, hipStream_t s) {
  if (!ws) return L2S_EINVAL;
  L2S_LAUNCH(maskpred_bwd_reduce_kernel, dim3(cdiv(C + 1, 256)), dim3(256), 0, s, ws, labels, num_fg, fg_max, C, MPB_CHUNKS, dw, db);
  return l2s_check_launch();
}
extern "C" int l2s_rcnn_predict(const float* heads, int ldh, int R, int ncls, const float* stds4, const float* means4, float* cls_prob,
                                float* bbox_pred, hipStream_t s) {
  L2S_LAUNCH(rcnn_predict_kernel, dim3(cdiv(R, 4)), dim3(256), 0, s, heads, ldh, R, ncls, stds4, means4, cls_prob, bbox_pred);
  return l2s_check_launch();
}
extern "C" int l2s_mask_prob(const float* score, int ldsc, int ncls, const int* labels, int ms2, long n_elem, float* out, hipStream_t s) {
  const long total = labels ? n_elem : n_elem * ncls;
  long g = (total + 255) / 256; if (g > 4096) g = 4096;
  L2S_LAUNCH(mask_prob_kernel, dim3((int)g), dim3(256), 0, s, score, ldsc, ncls, labels, ms2, n_elem, out);
  return l2s_check_launch();
}
