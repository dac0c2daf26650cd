// Bandwidth-bound helpers of the lang2seg train step (gfx950): shadow-weight generation, stem conv,
// pooling, elementwise glue and the fused SGD-momentum update.  All are HBM-bound byte movers: coalesced
// channel-contiguous (NHWC) accesses, grid-stride loops capped at ~8 blocks/CU.
#include "common.h"
#include "../../include/lang2seg_hip.h"

namespace {

__device__ __forceinline__ uint64_t mix64(uint64_t z) {  // splitmix64 finaliser
  z += 0x9E3779B97F4A7C15ull;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

inline int grid_for(long n, int block = 256, int cap = 2048) {
  long g = (n + block - 1) / block;
  if (g < 1) g = 1;
  return (int)(g > cap ? cap : g);
}

__global__ void weight_cast_kernel(const float* __restrict__ src, const float* __restrict__ scale, void* dst,
                                   long per_row, long total, int dt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    float v = src[i];
    if (scale) v *= scale[i / per_row];
    stx(dst, i, dt, v);
  }
}

// src [Cout][taps][Cin] -> dst [Cin][taps(reversed)][Cout]; 32x32 LDS tile transpose per tap
__global__ void weight_transpose_kernel(const float* __restrict__ src, const float* __restrict__ scale, void* dst,
                                        int Cout, int taps, int Cin, int dt) {
  __shared__ float tile[32][33];
  const int tap = blockIdx.z;
  const int co0 = blockIdx.y * 32, ci0 = blockIdx.x * 32;
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;  // 32 x 8
  for (int r = ty; r < 32; r += 8) {
    int co = co0 + r, ci = ci0 + tx;
    float v = 0.f;
    if (co < Cout && ci < Cin) { v = src[((long)co * taps + tap) * Cin + ci]; if (scale) v *= scale[co]; }
    tile[r][tx] = v;
  }
  __syncthreads();
  const int otap = taps - 1 - tap;
  for (int r = ty; r < 32; r += 8) {
    int ci = ci0 + r, co = co0 + tx;
    if (co < Cout && ci < Cin) stx(dst, ((long)ci * taps + otap) * Cout + co, dt, tile[tx][r]);
  }
}

// batched variant: one launch rebuilds every data-gradient weight copy of the step (descriptor table in device memory)
__global__ __launch_bounds__(256) void weight_transpose_batched_kernel(const l2s_transpose_desc* __restrict__ table, int n, int total, int dt_in) {
  // 64 (co) x 64 (ci) tiles through LDS: 16-byte reads along ci, 16-byte (bf16 x 8) / 32-byte (f32 x 8) writes along co.
  // One flat tile index over all descriptors (a workgroup strides through it and walks the table alongside): a fixed number of
  // workgroups per descriptor left most of them idle behind the few 3x3 layers (0.31 ms per step for 0.28 GB).
  __shared__ float tile[64][65];
  const int tid = threadIdx.x;
  int di = 0, base = 0;
  l2s_transpose_desc d = table[0];
  int tci = (d.Cin + 63) / 64, tco = (d.Cout + 63) / 64, ntiles = tci * tco * d.taps;
  for (int tt = blockIdx.x; tt < total; tt += gridDim.x) {
    while (tt >= base + ntiles && di + 1 < n) {
      base += ntiles; ++di;
      d = table[di];
      tci = (d.Cin + 63) / 64; tco = (d.Cout + 63) / 64; ntiles = tci * tco * d.taps;
    }
    const int t = tt - base;
    const int dt = (d.force_f32 & 1) ? L2S_F32 : dt_in;
    const bool src16 = d.force_f32 & 2;                  // src is the bf16 shadow (already scaled): half the bytes of the f32 master
    const bool vec_in = (d.Cin & 3) == 0, vec_out = (d.Cout & 7) == 0;
    const int tap = t / (tci * tco), rem = t - tap * (tci * tco);
    const int co0 = (rem / tci) * 64, ci0 = (rem % tci) * 64;
    __syncthreads();
    {
      const int c4 = (tid & 15) * 4;
#pragma unroll
      for (int pass = 0; pass < 4; ++pass) {
        const int r = (tid >> 4) + pass * 16;
        const int co = co0 + r, ci = ci0 + c4;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (co < d.Cout) {
          const float* sp = d.src + ((long)co * d.taps + tap) * d.Cin + ci;
          if (src16) {
            const bf16_t* hp = (const bf16_t*)d.src + ((long)co * d.taps + tap) * d.Cin + ci;
            if (vec_in && ci + 3 < d.Cin) { const uint2 u = *(const uint2*)hp; v = make_float4(bf2f(u.x & 0xffffu), bf2f(u.x >> 16), bf2f(u.y & 0xffffu), bf2f(u.y >> 16)); }
            else { if (ci < d.Cin) v.x = bf2f(hp[0]); if (ci + 1 < d.Cin) v.y = bf2f(hp[1]); if (ci + 2 < d.Cin) v.z = bf2f(hp[2]); if (ci + 3 < d.Cin) v.w = bf2f(hp[3]); }
          } else
          if (vec_in && ci + 3 < d.Cin) v = *(const float4*)sp;
          else { if (ci < d.Cin) v.x = sp[0]; if (ci + 1 < d.Cin) v.y = sp[1]; if (ci + 2 < d.Cin) v.z = sp[2]; if (ci + 3 < d.Cin) v.w = sp[3]; }
          if (d.scale) { const float sc = d.scale[co]; v.x *= sc; v.y *= sc; v.z *= sc; v.w *= sc; }
        }
        tile[r][c4] = v.x; tile[r][c4 + 1] = v.y; tile[r][c4 + 2] = v.z; tile[r][c4 + 3] = v.w;
      }
    }
    __syncthreads();
    const int otap = d.taps - 1 - tap;
    const int cg = (tid & 7) * 8;
#pragma unroll
    for (int pass = 0; pass < 2; ++pass) {
      const int r = (tid >> 3) + pass * 32;
      const int ci = ci0 + r, co = co0 + cg;
      if (ci >= d.Cin || co >= d.Cout) continue;
      float v[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) v[e] = tile[cg + e][r];
      const long o = ((long)ci * d.taps + otap) * d.Cout + co;
      if (vec_out && co + 7 < d.Cout) {
        if (dt == L2S_BF16) {
          uint4 pk;
          pk.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); pk.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
          pk.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); pk.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
          *(uint4*)((bf16_t*)d.dst + o) = pk;
        } else {
          *(float4*)((float*)d.dst + o) = make_float4(v[0], v[1], v[2], v[3]);
          *(float4*)((float*)d.dst + o + 4) = make_float4(v[4], v[5], v[6], v[7]);
        }
      } else {
        for (int e = 0; e < 8 && co + e < d.Cout; ++e) stx(d.dst, o + e, dt, v[e]);
      }
    }
  }
}

// Column sums (bias gradients), no atomics: a workgroup owns 64 columns and a range of rows, walked with 16 row lanes (coalesced
// 128/256-byte row pieces); the 16 partial sums are added in lane order.  One row range: out[c] += sum directly.  Tall matrices
// (12 544 x 256 for the mask head) are cut into row ranges whose partial sums go to part[range][cols]; colsum_finish adds them in order.
__global__ __launch_bounds__(1024) void colsum_kernel(const void* a, int rows, int cols, int lda, float* out, float* part, int rows_per, int dt) {
  __shared__ float sh[16][64];
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rg = threadIdx.x >> 6;
  const int r0 = blockIdx.y * rows_per, r1 = min(rows, r0 + rows_per);
  float s = 0.f;
  if (c < cols) {
    int r = r0 + rg;
    for (; r + 48 < r1; r += 64) {                       // four independent loads in flight
      const float v0 = ldx(a, (long)r * lda + c, dt), v1 = ldx(a, (long)(r + 16) * lda + c, dt);
      const float v2 = ldx(a, (long)(r + 32) * lda + c, dt), v3 = ldx(a, (long)(r + 48) * lda + c, dt);
      s += (v0 + v1) + (v2 + v3);
    }
    for (; r < r1; r += 16) s += ldx(a, (long)r * lda + c, dt);
  }
  sh[rg][threadIdx.x & 63] = s;
  __syncthreads();
  if (rg == 0 && c < cols) {
    float v = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) v += sh[g][threadIdx.x];
    if (part) part[(long)blockIdx.y * cols + c] = v; else out[c] += v;
  }
}
__global__ void colsum_finish_kernel(const float* part, int nr, int cols, float* out) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= cols) return;
  float v = 0.f;
  for (int q = 0; q < nr; ++q) v += part[(long)q * cols + c];
  out[c] += v;
}

// stem: one thread = one output pixel x 16 output channels; weights [64][7][7][3] staged in LDS as [tap*3+c][64]
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ w,
                                                  const float* __restrict__ scale, const float* __restrict__ bias, void* y,
                                                  int H, int W, int OH, int OW, int dt) {
  __shared__ float ws[147 * 64];
  for (int i = threadIdx.x; i < 147 * 64; i += 256) { int k = i >> 6, co = i & 63; ws[i] = w[co * 147 + k]; }
  __syncthreads();
  const int cg = threadIdx.x & 3;                       // 4 channel groups of 16
  const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pix >= (long)OH * OW) return;
  const int oy = (int)(pix / OW), ox = (int)(pix - (long)oy * OW);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int ky = 0; ky < 7; ++ky) {
    int iy = oy * 2 - 3 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 7; ++kx) {
      int ix = ox * 2 - 3 + kx;
      if (ix < 0 || ix >= W) continue;
      const float* px = img + ((long)iy * W + ix) * 3;
      const float* wk = ws + ((ky * 7 + kx) * 3) * 64 + cg * 16;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        float v = px[c];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(v, wk[c * 64 + i], acc[i]);
      }
    }
  }
#pragma unroll
  for (int i = 0; i < 16; ++i) {
    int co = cg * 16 + i;
    stx(y, pix * 64 + co, dt, fmaxf(acc[i] * scale[co] + bias[co], 0.f));
  }
}


// ---- VGG16 variant (nets/vgg16.py:43-54): first convolution (3 -> 64, 3x3, pad 1, bias, ReLU; frozen, forward only) on the
// fp32 NHWC image, same layout trick as the stem kernel (weights [64][3][3][3] staged as [tap*3+c][64]) ----
__global__ __launch_bounds__(256) void conv3x3_c3_kernel(const float* __restrict__ img, const float* __restrict__ w, const float* __restrict__ bias,
                                                        void* y, int H, int W, int dt) {
  __shared__ float ws[27 * 64];
  for (int i = threadIdx.x; i < 27 * 64; i += 256) { int k = i >> 6, co = i & 63; ws[i] = w[co * 27 + k]; }
  __syncthreads();
  const int cg = threadIdx.x & 3;
  const long pix = (long)blockIdx.x * 64 + (threadIdx.x >> 2);
  if (pix >= (long)H * W) return;
  const int oy = (int)(pix / W), ox = (int)(pix - (long)oy * W);
  float acc[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = oy - 1 + ky;
    if (iy < 0 || iy >= H) continue;
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = ox - 1 + kx;
      if (ix < 0 || ix >= W) continue;
      const float* px = img + ((long)iy * W + ix) * 3;
      const float* wk = ws + ((ky * 3 + kx) * 3) * 64 + cg * 16;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float v = px[c];
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[i] = fmaf(v, wk[c * 64 + i], acc[i]);
      }
    }
  }
  // (round 5: the thread's 16 channels leave as two 16-byte stores in bf16 / four in f32 - sixteen 2-byte stores per thread made this launch 0.23 ms
  // at 600x1000, 0.004 of peak and ten times its HBM floor)
  float v[16];
#pragma unroll
  for (int i = 0; i < 16; ++i) v[i] = fmaxf(acc[i] + bias[cg * 16 + i], 0.f);
  if (dt) {
    uint32_t pk[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) pk[i] = (uint32_t)f2bf(v[2 * i]) | ((uint32_t)f2bf(v[2 * i + 1]) << 16);
    uint4* dst = (uint4*)((bf16_t*)y + pix * 64 + cg * 16);
    dst[0] = make_uint4(pk[0], pk[1], pk[2], pk[3]); dst[1] = make_uint4(pk[4], pk[5], pk[6], pk[7]);
  } else {
    float4* dst = (float4*)((float*)y + pix * 64 + cg * 16);
#pragma unroll
    for (int i = 0; i < 4; ++i) dst[i] = make_float4(v[4 * i], v[4 * i + 1], v[4 * i + 2], v[4 * i + 3]);
  }
}
// 2x2 / stride 2 max pooling, floor mode (nn.MaxPool2d(2, 2)), NHWC, n_img images
__global__ void maxpool2x2_fwd_kernel(const void* x, void* y, int n_img, int IH, int IW, int C, int OH, int OW, int dt) {
  const long total = (long)n_img * OH * OW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long p = i / C; const int ox = (int)(p % OW); p /= OW; const int oy = (int)(p % OH); const long n = p / OH;
    const long b = ((n * IH + 2 * oy) * IW + 2 * ox) * C + c;
    const float m = fmaxf(fmaxf(ldx(x, b, dt), ldx(x, b + C, dt)), fmaxf(ldx(x, b + (long)IW * C, dt), ldx(x, b + (long)IW * C + C, dt)));
    stx(y, i, dt, m);
  }
}
// backward: the whole gradient goes to the first maximum of the window in (row, column) order (ATen max_pool2d backward);
// relu != 0: x is a ReLU output and dx is the gradient w.r.t. its pre-activation (zero where the maximum is 0).
// Rows / columns not covered by a window (odd sizes) get 0.
__global__ void maxpool2x2_bwd_kernel(const void* dy, const void* x, void* dx, int n_img, int IH, int IW, int C, int OH, int OW, int dt, int relu) {
  const int WH = (IH + 1) / 2, WW = (IW + 1) / 2;         // windows incl. the partial ones at the odd border
  const long total = (long)n_img * WH * WW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C); long p = i / C; const int wx = (int)(p % WW); p /= WW; const int wy = (int)(p % WH); const long n = p / WH;
    const long b = ((n * IH + 2 * wy) * IW + 2 * wx) * C + c;
    const bool full = wy < OH && wx < OW;
    if (!full) {                                           // border strip outside every pooling window
      stx(dx, b, dt, 0.f);
      if (2 * wx + 1 < IW) stx(dx, b + C, dt, 0.f);
      if (2 * wy + 1 < IH) { stx(dx, b + (long)IW * C, dt, 0.f); if (2 * wx + 1 < IW) stx(dx, b + (long)IW * C + C, dt, 0.f); }
      continue;
    }
    const long o[4] = {b, b + C, b + (long)IW * C, b + (long)IW * C + C};
    float m = ldx(x, o[0], dt); int am = 0;
#pragma unroll
    for (int k = 1; k < 4; ++k) { const float v = ldx(x, o[k], dt); if (v > m) { m = v; am = k; } }
    float g = ldx(dy, ((n * OH + wy) * OW + wx) * C + c, dt);
    if (relu && !(m > 0.f)) g = 0.f;
#pragma unroll
    for (int k = 0; k < 4; ++k) stx(dx, o[k], dt, k == am ? g : 0.f);
  }
}

__global__ void maxpool_kernel(const void* x, void* y, int IH, int IW, int C, int OH, int OW, int dt) {
  const long total = (long)OH * OW * C;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long p = i / C; int ox = (int)(p % OW), oy = (int)(p / OW);
    float m = -INFINITY;
    for (int ky = 0; ky < 3; ++ky) {
      int iy = oy * 2 - 1 + ky; if (iy < 0 || iy >= IH) continue;
      for (int kx = 0; kx < 3; ++kx) {
        int ix = ox * 2 - 1 + kx; if (ix < 0 || ix >= IW) continue;
        m = fmaxf(m, ldx(x, ((long)iy * IW + ix) * C + c, dt));
      }
    }
    stx(y, i, dt, m);
  }
}

__global__ void fill_kernel(float* p, float v, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) p[i] = v;
}
__global__ void cast_kernel(const void* s, int sd, void* d, int dd, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) stx(d, i, dd, ldx(s, i, sd));
}
__global__ void mul_kernel(const float* a, const float* b, float* o, long n) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] * b[i];
}
// out = a + b, 16 bytes per lane (n % 4 == 0, 16-byte aligned) with a scalar tail otherwise
__global__ void add_f32_kernel(const float* a, const float* b, float* o, long n) {
  const long n4 = (((uintptr_t)a | (uintptr_t)b | (uintptr_t)o) & 15) ? 0 : n / 4;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n4; i += (long)gridDim.x * blockDim.x) {
    const float4 x = ((const float4*)a)[i], y = ((const float4*)b)[i];
    ((float4*)o)[i] = make_float4(x.x + y.x, x.y + y.y, x.z + y.z, x.w + y.w);
  }
  for (long i = n4 * 4 + blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}
// out = x * mask (fp32 dropout mask), zeroed where relu_ref <= 0 when relu_ref is given (dropout forward / backward of fc6, fc7)
__global__ void scale_mask_kernel(const void* x, const float* mask, const void* ref, void* out, long n, int dt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = ldx(x, i, dt) * mask[i];
    if (ref && !(ldx(ref, i, dt) > 0.f)) v = 0.f;
    stx(out, i, dt, v);
  }
}
__global__ void add3_kernel(const void* a, const void* b, const float* c, void* d, long n, int dt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = ldx(a, i, dt);
    if (b) v += ldx(b, i, dt);
    if (c) v += c[i];
    stx(d, i, dt, v);
  }
}

// bf16 x 8 forms (16-byte accesses) of the elementwise kernels on the main queue
__global__ void add3_bf16x8_kernel(const bf16_t* __restrict__ a, const bf16_t* __restrict__ b, const float* __restrict__ c, bf16_t* d, long n8) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n8; i += (long)gridDim.x * blockDim.x) {
    const uint4 qa = *(const uint4*)(a + i * 8);
    const uint32_t aw[4] = {qa.x, qa.y, qa.z, qa.w};
    float v[8];
#pragma unroll
    for (int e = 0; e < 4; ++e) { v[2 * e] = __uint_as_float(aw[e] << 16); v[2 * e + 1] = __uint_as_float(aw[e] & 0xFFFF0000u); }
    if (b) {
      const uint4 qb = *(const uint4*)(b + i * 8);
      const uint32_t bw[4] = {qb.x, qb.y, qb.z, qb.w};
#pragma unroll
      for (int e = 0; e < 4; ++e) { v[2 * e] += __uint_as_float(bw[e] << 16); v[2 * e + 1] += __uint_as_float(bw[e] & 0xFFFF0000u); }
    }
    if (c) {
      const float4 c0 = *(const float4*)(c + i * 8), c1 = *(const float4*)(c + i * 8 + 4);
      v[0] += c0.x; v[1] += c0.y; v[2] += c0.z; v[3] += c0.w; v[4] += c1.x; v[5] += c1.y; v[6] += c1.z; v[7] += c1.w;
    }
    uint4 o;
    o.x = (uint32_t)f2bf(v[0]) | ((uint32_t)f2bf(v[1]) << 16); o.y = (uint32_t)f2bf(v[2]) | ((uint32_t)f2bf(v[3]) << 16);
    o.z = (uint32_t)f2bf(v[4]) | ((uint32_t)f2bf(v[5]) << 16); o.w = (uint32_t)f2bf(v[6]) | ((uint32_t)f2bf(v[7]) << 16);
    *(uint4*)(d + i * 8) = o;
  }
}
__global__ void avgpool_fwd_bf16x8_kernel(const bf16_t* __restrict__ x, bf16_t* y, int hw, int C) {
  const int n = blockIdx.y;
  const int c = (blockIdx.x * blockDim.x + threadIdx.x) * 8;
  if (c >= C) return;
  float s[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) s[e] = 0.f;
  for (int p = 0; p < hw; ++p) {
    const uint4 q = *(const uint4*)(x + ((long)n * hw + p) * C + c);
    const uint32_t qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
    for (int e = 0; e < 4; ++e) { s[2 * e] += __uint_as_float(qw[e] << 16); s[2 * e + 1] += __uint_as_float(qw[e] & 0xFFFF0000u); }
  }
  uint4 o;
  const float inv = (float)hw;
  o.x = (uint32_t)f2bf(s[0] / inv) | ((uint32_t)f2bf(s[1] / inv) << 16); o.y = (uint32_t)f2bf(s[2] / inv) | ((uint32_t)f2bf(s[3] / inv) << 16);
  o.z = (uint32_t)f2bf(s[4] / inv) | ((uint32_t)f2bf(s[5] / inv) << 16); o.w = (uint32_t)f2bf(s[6] / inv) | ((uint32_t)f2bf(s[7] / inv) << 16);
  *(uint4*)(y + (long)n * C + c) = o;
}

__global__ void avgpool_fwd_kernel(const void* x, void* y, int hw, int C, int dt) {
  const int n = blockIdx.y;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  float s = 0.f;
  for (int p = 0; p < hw; ++p) s += ldx(x, ((long)n * hw + p) * C + c, dt);
  // reference: .mean(3).mean(2) -> mean over W then mean over H; for hw = P*P both are sums / P / P
  stx(y, (long)n * C + c, dt, s / (float)hw);
}
__global__ void avgpool_bwd_kernel(const void* dy, void* dx, const void* addend, const void* ref, int hw, int C, long total, int dt) {
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    int c = (int)(i % C); long n = i / ((long)hw * C);
    float v = ldx(dy, n * C + c, dt) / (float)hw;
    if (addend) v += ldx(addend, i, dt);
    if (ref && !(ldx(ref, i, dt) > 0.f)) v = 0.f;
    stx(dx, i, dt, v);
  }
}
// bf16, C % 8 == 0: 8 channels (16 bytes) per thread
__global__ void avgpool_bwd_bf16x8_kernel(const bf16_t* dy, bf16_t* dx, const bf16_t* addend, const bf16_t* ref, int hw, int C, long total8) {
  const float inv = 1.f / (float)hw;
  const int C8 = C >> 3;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < total8; i += (long)gridDim.x * blockDim.x) {
    const int c8 = (int)(i % C8); const long n = i / ((long)hw * C8);
    const uint4 g = *(const uint4*)(dy + n * C + (long)c8 * 8);
    uint4 a = make_uint4(0, 0, 0, 0), r = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
    if (addend) a = *(const uint4*)(addend + i * 8);
    if (ref) r = *(const uint4*)(ref + i * 8);
    const uint32_t gw[4] = {g.x, g.y, g.z, g.w}, aw[4] = {a.x, a.y, a.z, a.w}, rw[4] = {r.x, r.y, r.z, r.w};
    uint32_t ow[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      float v0 = __uint_as_float(gw[k] << 16) * inv, v1 = __uint_as_float(gw[k] & 0xFFFF0000u) * inv;
      if (addend) { v0 += __uint_as_float(aw[k] << 16); v1 += __uint_as_float(aw[k] & 0xFFFF0000u); }
      if (!(__uint_as_float(rw[k] << 16) > 0.f)) v0 = 0.f;
      if (!(__uint_as_float(rw[k] & 0xFFFF0000u) > 0.f)) v1 = 0.f;
      ow[k] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
    }
    *(uint4*)(dx + i * 8) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
  }
}

__device__ __forceinline__ int bin_lo(int i, int n_in, int n_out) { return (i * n_in) / n_out; }
__device__ __forceinline__ int bin_hi(int i, int n_in, int n_out) { return ((i + 1) * n_in + n_out - 1) / n_out; }

__global__ void adaptive_pool_fwd_kernel(const void* x, const float* pm, void* y, int H, int W, int C, int OH, int OW, int ldy, int dt) {
  const int bin = blockIdx.y, oy = bin / OW, ox = bin - oy * OW;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const int y0 = bin_lo(oy, H, OH), y1 = bin_hi(oy, H, OH), x0 = bin_lo(ox, W, OW), x1 = bin_hi(ox, W, OW);
  float s = 0.f;
  for (int yy = y0; yy < y1; ++yy)
    for (int xx = x0; xx < x1; ++xx) {
      float v = ldx(x, ((long)yy * W + xx) * C + c, dt);
      if (pm) v *= pm[yy * W + xx];
      s += v;
    }
  stx(y, (long)bin * ldy + c, dt, s / (float)((y1 - y0) * (x1 - x0)));
}

__global__ void adaptive_pool_bwd_kernel(const void* dy, int lddy, int off_all, int off_mask, const float* pm, void* dx,
                                         const void* ref, int H, int W, int C, int OH, int OW, int dt) {
  const int pix = blockIdx.y, yy = pix / W, xx = pix - yy * W;
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  // bins whose [lo,hi) range contains the pixel: only the bins around floor(yy OH / H) can (bins overlap by at most one pixel)
  float g = 0.f;
  const float m = pm ? pm[pix] : 0.f;
  const int oyc = (yy * OH) / H, oxc = (xx * OW) / W;
  const int oya = OH <= H ? max(0, oyc - 1) : 0, oyb = OH <= H ? min(OH - 1, oyc + 1) : OH - 1;
  const int oxa = OW <= W ? max(0, oxc - 1) : 0, oxb = OW <= W ? min(OW - 1, oxc + 1) : OW - 1;
  for (int oy = oya; oy <= oyb; ++oy) {
    int y0 = bin_lo(oy, H, OH), y1 = bin_hi(oy, H, OH);
    if (yy < y0 || yy >= y1) continue;
    for (int ox = oxa; ox <= oxb; ++ox) {
      int x0 = bin_lo(ox, W, OW), x1 = bin_hi(ox, W, OW);
      if (xx < x0 || xx >= x1) continue;
      float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
      long b = (long)(oy * OW + ox) * lddy;
      float d = ldx(dy, b + off_all + c, dt);
      if (pm) d += m * ldx(dy, b + off_mask + c, dt);
      g += d * inv;
    }
  }
  long o = (long)pix * C + c;
  if (ref && !(ldx(ref, o, dt) > 0.f)) g = 0.f;
  stx(dx, o, dt, g);
}

// bf16, C % 8 == 0 and 8-element-aligned dy columns: 8 channels (16 bytes) per thread, one pixel per blockIdx.x (the bin search is
// uniform per workgroup: scalar).  The one-channel-per-thread kernel above took 80 us for the 2394 x 2048 caption-branch map.
__global__ __launch_bounds__(256) void adaptive_pool_bwd_bf16x8_kernel(const bf16_t* __restrict__ dy, int lddy, int off_all, int off_mask, const float* __restrict__ pm,
                                                                      bf16_t* dx, const bf16_t* __restrict__ ref, int H, int W, int C, int OH, int OW) {
  const int pix = blockIdx.x, yy = pix / W, xx = pix - yy * W;
  const int c = (blockIdx.y * blockDim.x + threadIdx.x) * 8;
  if (c >= C) return;
  float g[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) g[k] = 0.f;
  const float m = pm ? pm[pix] : 0.f;
  const int oyc = (yy * OH) / H, oxc = (xx * OW) / W;
  const int oya = OH <= H ? max(0, oyc - 1) : 0, oyb = OH <= H ? min(OH - 1, oyc + 1) : OH - 1;
  const int oxa = OW <= W ? max(0, oxc - 1) : 0, oxb = OW <= W ? min(OW - 1, oxc + 1) : OW - 1;
  for (int oy = oya; oy <= oyb; ++oy) {
    const int y0 = bin_lo(oy, H, OH), y1 = bin_hi(oy, H, OH);
    if (yy < y0 || yy >= y1) continue;
    for (int ox = oxa; ox <= oxb; ++ox) {
      const int x0 = bin_lo(ox, W, OW), x1 = bin_hi(ox, W, OW);
      if (xx < x0 || xx >= x1) continue;
      const float inv = 1.f / (float)((y1 - y0) * (x1 - x0));
      const long b = (long)(oy * OW + ox) * lddy;
      const uint4 a = *(const uint4*)(dy + b + off_all + c);
      uint4 q = make_uint4(0, 0, 0, 0);
      if (pm) q = *(const uint4*)(dy + b + off_mask + c);
      const uint32_t aw[4] = {a.x, a.y, a.z, a.w}, qw[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        float d0 = __uint_as_float(aw[k] << 16), d1 = __uint_as_float(aw[k] & 0xFFFF0000u);
        if (pm) { d0 += m * __uint_as_float(qw[k] << 16); d1 += m * __uint_as_float(qw[k] & 0xFFFF0000u); }
        g[2 * k] += d0 * inv; g[2 * k + 1] += d1 * inv;
      }
    }
  }
  const long o = (long)pix * C + c;
  uint4 r = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
  if (ref) r = *(const uint4*)(ref + o);
  const uint32_t rw[4] = {r.x, r.y, r.z, r.w};
  uint32_t ow[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    float v0 = g[2 * k], v1 = g[2 * k + 1];
    if (!(__uint_as_float(rw[k] << 16) > 0.f)) v0 = 0.f;
    if (!(__uint_as_float(rw[k] & 0xFFFF0000u) > 0.f)) v1 = 0.f;
    ow[k] = (uint32_t)f2bf(v0) | ((uint32_t)f2bf(v1) << 16);
  }
  *(uint4*)(dx + o) = make_uint4(ow[0], ow[1], ow[2], ow[3]);
}

__global__ void mask_downsample_kernel(const uint8_t* mask, float* out, int H, int W, int h, int w) {
  // one wave per output pixel
  const int o = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (o >= h * w) return;
  const int oy = o / w, ox = o - oy * w;
  const int y0 = bin_lo(oy, H, h), y1 = bin_hi(oy, H, h), x0 = bin_lo(ox, W, w), x1 = bin_hi(ox, W, w);
  const int bw = x1 - x0, n = (y1 - y0) * bw;
  float s = 0.f;
  for (int i = lane; i < n; i += 64) { int yy = y0 + i / bw, xx = x0 + i % bw; s += (float)mask[(long)yy * W + xx]; }
  s = wave_sum(s);
  if (lane == 0) out[o] = (s / (float)n >= 0.5f) ? 1.f : 0.f;
}

__global__ void counter_inc_kernel(uint64_t* c) { *c += 1; }
__global__ void stamp_kernel(uint64_t* out) { *out = wall_clock64(); }   // constant 100 MHz device clock
__global__ void dropout_kernel(float* m, long n, float p, const uint64_t* seed_dev, uint64_t salt) {
  const uint64_t seed = mix64(*seed_dev * 0x9E3779B97F4A7C15ull + salt);
  const float keep = 1.f - p, inv = keep > 0.f ? 1.f / keep : 0.f;
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    uint32_t r = (uint32_t)(mix64(seed * 0x100000001B3ull + (uint64_t)i) >> 40);  // 24 bits
    m[i] = ((float)r * (1.f / 16777216.f) < keep) ? inv : 0.f;
  }
}
__global__ void keys_kernel(uint32_t* k, long n, const uint64_t* seed_dev, uint64_t salt) {
  const uint64_t seed = mix64(*seed_dev * 0x9E3779B97F4A7C15ull + salt);
  for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    k[i] = (uint32_t)(mix64(seed * 0x100000001B3ull + (uint64_t)i) >> 32);
}

// SGD with momentum over a table of segments (one per tensor), fused with the dtype shadow rewrite and (clear != 0) the gradient clear.
// Persistent grid: the work is cut into chunks of SGD_CHUNK elements of one segment (l2s_sgd_seg.chunk0 = index of the segment's first
// chunk, filled by the host); a workgroup walks chunks blockIdx.x, blockIdx.x + gridDim.x, ...  A thread owns four 16-byte vectors of
// each operand and requests all twelve before the first use, so one resident workgroup per CU already keeps ~48 KB in flight (the old
// form launched 64 workgroups per tensor, 10 000 short workgroups that held every wave slot of the chip while the next step's frozen
// prefix waited).
constexpr int SGD_CHUNK = 4096;
__device__ __forceinline__ void sgd_elem(float g, float& w, float& m, float rs, float gscale, float lwd, float momentum, float llr) {
  const float gg = g * gscale * rs + lwd * w;
  m = momentum * m + gg;
  w = w - llr * m;
}
// flags: 1 = zero the gradient as it is read; 2 = only rewrite the dtype shadow from the parameters (no update: after an all-gather of
// weights other ranks updated).  [lo, hi): only elements at these offsets of the flat buffer are touched (a rank's shard of a gradient
// bucket); [chunk_lo, chunk_hi): the chunks of the table that can hold them (numbered from the table's first chunk).
__global__ __launch_bounds__(256) void sgd_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ mom,
                                                  const l2s_sgd_seg* __restrict__ segs, int nseg, const float* __restrict__ rowscale,
                                                  float lr, float momentum, float wd, float gscale, void* shadow, int sdt, int flags,
                                                  long lo, long hi, int chunk_lo, int chunk_hi, const bf16_t* __restrict__ g16, long g16_lo) {
  const bool clear = flags & 1, shadow_only = flags & 2;
  const int c0 = segs[0].chunk0;
  int total = segs[nseg - 1].chunk0 - c0 + (int)((segs[nseg - 1].count + SGD_CHUNK - 1) / SGD_CHUNK);
  if (chunk_hi >= 0 && chunk_hi < total) total = chunk_hi;
  for (int c = (chunk_lo > 0 ? chunk_lo : 0) + blockIdx.x; c < total; c += gridDim.x) {
    int slo = 0, shi = nseg - 1;
    while (slo < shi) {                                 // the segment that holds chunk c (uniform: scalar loads)
      const int mid = (slo + shi + 1) >> 1;
      if (segs[mid].chunk0 - c0 <= c) slo = mid; else shi = mid - 1;
    }
    const l2s_sgd_seg sg = segs[slo];
    const long base = (long)(c - (sg.chunk0 - c0)) * SGD_CHUNK;
    const int n = (int)((sg.count - base) < SGD_CHUNK ? (sg.count - base) : SGD_CHUNK);
    const float lwd = sg.weight_decay ? wd : 0.f;
    const float llr = lr * sg.lr_mult;
    const bool clr = clear && !(sg.flags & 1);          // (a gradient its producer overwrites whole needs no clear)
    const long o0 = sg.offset + base;
    const bool inside = o0 >= lo && o0 + n <= hi;       // (a chunk that the range cuts goes element by element)
    const bool vec = inside && ((sg.offset & 3) == 0) && (sg.rowscale_off < 0 || (sg.row_len & 3) == 0);
    const int nv = vec ? (n >> 2) : 0;
    float4 g4[4], w4[4], m4[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = threadIdx.x + j * 256;
      if (v < nv) {
        w4[j] = *(const float4*)(param + o0 + 4 * v);
        if (!shadow_only) {
          if (g16) {                                   // the reduce-scattered bf16 shard itself (data parallel): no cast pass back into the f32 gradient buffer
            const uint2 pk = *(const uint2*)(g16 + (o0 + 4 * v - g16_lo));
            g4[j] = make_float4(bf2f((bf16_t)(pk.x & 0xffffu)), bf2f((bf16_t)(pk.x >> 16)), bf2f((bf16_t)(pk.y & 0xffffu)), bf2f((bf16_t)(pk.y >> 16)));
          } else g4[j] = *(const float4*)(grad + o0 + 4 * v);
          m4[j] = *(const float4*)(mom + o0 + 4 * v);
        }
      }
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int v = threadIdx.x + j * 256;
      if (v >= nv) continue;
      const long o = o0 + 4 * v;
      const float rs = sg.rowscale_off >= 0 ? rowscale[sg.rowscale_off + (unsigned)(base + 4 * v) / (unsigned)sg.row_len] : 1.f;
      float ww[4] = {w4[j].x, w4[j].y, w4[j].z, w4[j].w};
      if (!shadow_only) {
        float gg[4] = {g4[j].x, g4[j].y, g4[j].z, g4[j].w}, mm[4] = {m4[j].x, m4[j].y, m4[j].z, m4[j].w};
#pragma unroll
        for (int e = 0; e < 4; ++e) sgd_elem(gg[e], ww[e], mm[e], rs, gscale, lwd, momentum, llr);
        *(float4*)(mom + o) = make_float4(mm[0], mm[1], mm[2], mm[3]);
        *(float4*)(param + o) = make_float4(ww[0], ww[1], ww[2], ww[3]);
        if (clr && !g16) *(float4*)(grad + o) = make_float4(0.f, 0.f, 0.f, 0.f);
      }
      if (shadow) {
        if (sdt) {
          uint2 pk; pk.x = (uint32_t)f2bf(ww[0] * rs) | ((uint32_t)f2bf(ww[1] * rs) << 16); pk.y = (uint32_t)f2bf(ww[2] * rs) | ((uint32_t)f2bf(ww[3] * rs) << 16);
          *(uint2*)((bf16_t*)shadow + o) = pk;
        } else *(float4*)((float*)shadow + o) = make_float4(ww[0] * rs, ww[1] * rs, ww[2] * rs, ww[3] * rs);
      }
    }
    for (int i = (nv << 2) + threadIdx.x; i < n; i += 256) {
      const long o = o0 + i;
      if (o < lo || o >= hi) continue;
      const float rs = sg.rowscale_off >= 0 ? rowscale[sg.rowscale_off + (unsigned)(base + i) / (unsigned)sg.row_len] : 1.f;
      float w = param[o];
      if (!shadow_only) {
        float m = mom[o];
        sgd_elem(g16 ? bf2f(g16[o - g16_lo]) : grad[o], w, m, rs, gscale, lwd, momentum, llr);
        mom[o] = m; param[o] = w;
        if (clr && !g16) grad[o] = 0.f;
      }
      if (shadow) stx(shadow, o, sdt, w * rs);
    }
  }
}

}  // namespace

extern "C" int l2s_version(void) { return 100; }

extern "C" int l2s_weight_cast(const float* src, const float* scale, void* dst, int Cout, int taps, int Cin, int dtype, hipStream_t s) {
  long per_row = (long)taps * Cin, total = per_row * Cout;
  L2S_LAUNCH(weight_cast_kernel, dim3(grid_for(total)), dim3(256), 0, s, src, scale, dst, per_row, total, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_weight_transpose(const float* src, const float* scale, void* dst, int Cout, int taps, int Cin, int dtype, hipStream_t s) {
  dim3 grid(cdiv(Cin, 32), cdiv(Cout, 32), taps);
  L2S_LAUNCH(weight_transpose_kernel, grid, dim3(256), 0, s, src, scale, dst, Cout, taps, Cin, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_weight_transpose_batched(const l2s_transpose_desc* table_dev, int n, int total_tiles, int dtype, hipStream_t s) {
  if (n <= 0 || total_tiles <= 0) return L2S_OK;
  L2S_LAUNCH(weight_transpose_batched_kernel, dim3(total_tiles < 2048 ? total_tiles : 2048), dim3(256), 0, s, table_dev, n, total_tiles, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_colsum(const void* a, int rows, int cols, int lda, float* out, float* ws, long ws_floats, int dtype, hipStream_t s) {
  int nr = (rows + 511) / 512;                           // row ranges of >= 512 rows
  if (nr > 32) nr = 32;
  if (!ws || ws_floats < (long)nr * cols) nr = 1;
  const int rows_per = (rows + nr - 1) / nr;
  float* part = nr > 1 ? ws : nullptr;
  L2S_LAUNCH(colsum_kernel, dim3(cdiv(cols, 64), nr), dim3(1024), 0, s, a, rows, cols, lda, out, part, rows_per, dtype);
  if (nr > 1) L2S_LAUNCH(colsum_finish_kernel, dim3(cdiv(cols, 256)), dim3(256), 0, s, (const float*)ws, nr, cols, out);
  return l2s_check_launch();
}
extern "C" int l2s_stem_conv(const float* img, const float* w, const float* scale, const float* bias, void* y, int H, int W,
                             int OH, int OW, int dtype, hipStream_t s) {
  L2S_LAUNCH(stem_kernel, dim3(cdiv((long)OH * OW, 64)), dim3(256), 0, s, img, w, scale, bias, y, H, W, OH, OW, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_conv3x3_c3(const float* img, const float* w, const float* bias, void* y, int H, int W, int dtype, hipStream_t s) {
  L2S_LAUNCH(conv3x3_c3_kernel, dim3(cdiv((long)H * W, 64)), dim3(256), 0, s, img, w, bias, y, H, W, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_maxpool2x2_fwd(const void* x, void* y, int n_img, int IH, int IW, int C, int dtype, hipStream_t s) {
  const int OH = IH / 2, OW = IW / 2;
  L2S_LAUNCH(maxpool2x2_fwd_kernel, dim3(grid_for((long)n_img * OH * OW * C)), dim3(256), 0, s, x, y, n_img, IH, IW, C, OH, OW, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_maxpool2x2_bwd(const void* dy, const void* x, void* dx, int n_img, int IH, int IW, int C, int relu_out, int dtype, hipStream_t s) {
  const int OH = IH / 2, OW = IW / 2;
  L2S_LAUNCH(maxpool2x2_bwd_kernel, dim3(grid_for((long)n_img * ((IH + 1) / 2) * ((IW + 1) / 2) * C)), dim3(256), 0, s, dy, x, dx, n_img, IH, IW, C, OH, OW,
             dtype, relu_out);
  return l2s_check_launch();
}
extern "C" int l2s_maxpool3x3s2(const void* x, void* y, int IH, int IW, int C, int OH, int OW, int dtype, hipStream_t s) {
  L2S_LAUNCH(maxpool_kernel, dim3(grid_for((long)OH * OW * C)), dim3(256), 0, s, x, y, IH, IW, C, OH, OW, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_fill_f32(float* p, float v, long n, hipStream_t s) {
  if (n <= 0) return L2S_OK;
  L2S_LAUNCH(fill_kernel, dim3(grid_for(n)), dim3(256), 0, s, p, v, n);
  return l2s_check_launch();
}
extern "C" int l2s_cast(const void* src, int sd, void* dst, int dd, long n, hipStream_t s) {
  L2S_LAUNCH(cast_kernel, dim3(grid_for(n)), dim3(256), 0, s, src, sd, dst, dd, n);
  return l2s_check_launch();
}
extern "C" int l2s_mul_f32(const float* a, const float* b, float* out, long n, hipStream_t s) {
  L2S_LAUNCH(mul_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, out, n);
  return l2s_check_launch();
}
extern "C" int l2s_add_f32(const float* a, const float* b, float* out, long n, hipStream_t s) {
  L2S_LAUNCH(add_f32_kernel, dim3(grid_for((n + 3) / 4)), dim3(256), 0, s, a, b, out, n);
  return l2s_check_launch();
}
extern "C" int l2s_scale_mask(const void* x, const float* mask, const void* relu_ref, void* out, long n, int dtype, hipStream_t s) {
  L2S_LAUNCH(scale_mask_kernel, dim3(grid_for(n)), dim3(256), 0, s, x, mask, relu_ref, out, n, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_add3(const void* a, const void* b, const float* c, void* dst, long n, int dtype, hipStream_t s) {
  if (dtype == L2S_BF16 && !(n & 7) && !((uintptr_t)a & 15) && !((uintptr_t)b & 15) && !((uintptr_t)c & 15) && !((uintptr_t)dst & 15)) {
    L2S_LAUNCH(add3_bf16x8_kernel, dim3(grid_for(n / 8)), dim3(256), 0, s, (const bf16_t*)a, (const bf16_t*)b, c, (bf16_t*)dst, n / 8);
    return l2s_check_launch();
  }
  L2S_LAUNCH(add3_kernel, dim3(grid_for(n)), dim3(256), 0, s, a, b, c, dst, n, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_avgpool_fwd(const void* x, void* y, int n_img, int hw, int C, int dtype, hipStream_t s) {
  if (dtype == L2S_BF16 && !(C & 7) && !((uintptr_t)x & 15) && !((uintptr_t)y & 15)) {
    const int th = C / 8 >= 256 ? 256 : ((C / 8 + 63) / 64) * 64;
    L2S_LAUNCH(avgpool_fwd_bf16x8_kernel, dim3(cdiv(C / 8, th), n_img), dim3(th), 0, s, (const bf16_t*)x, (bf16_t*)y, hw, C);
    return l2s_check_launch();
  }
  L2S_LAUNCH(avgpool_fwd_kernel, dim3(cdiv(C, 256), n_img), dim3(256), 0, s, x, y, hw, C, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_avgpool_bwd(const void* dy, void* dx, const void* addend, const void* ref, int n_img, int hw, int C, int dtype, hipStream_t s) {
  long total = (long)n_img * hw * C;
  if (dtype == L2S_BF16 && (C & 7) == 0) {
    L2S_LAUNCH(avgpool_bwd_bf16x8_kernel, dim3(grid_for(total / 8)), dim3(256), 0, s, (const bf16_t*)dy, (bf16_t*)dx, (const bf16_t*)addend, (const bf16_t*)ref, hw, C, total / 8);
    return l2s_check_launch();
  }
  L2S_LAUNCH(avgpool_bwd_kernel, dim3(grid_for(total)), dim3(256), 0, s, dy, dx, addend, ref, hw, C, total, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_adaptive_pool_fwd(const void* x, const float* pm, void* y, int H, int W, int C, int OH, int OW, int ldy, int dtype, hipStream_t s) {
  L2S_LAUNCH(adaptive_pool_fwd_kernel, dim3(cdiv(C, 256), OH * OW), dim3(256), 0, s, x, pm, y, H, W, C, OH, OW, ldy, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_adaptive_pool_bwd(const void* dy, int lddy, int off_all, int off_mask, const float* pm, void* dx, const void* ref,
                                     int H, int W, int C, int OH, int OW, int dtype, hipStream_t s) {
  if (dtype == L2S_BF16 && !(C & 7) && !(lddy & 7) && !(off_all & 7) && !(off_mask & 7) && !((uintptr_t)dy & 15) && !((uintptr_t)dx & 15) && !((uintptr_t)ref & 15)) {
    const int threads = C / 8 < 256 ? ((C / 8 + 63) / 64) * 64 : 256;
    L2S_LAUNCH(adaptive_pool_bwd_bf16x8_kernel, dim3(H * W, cdiv(C / 8, threads)), dim3(threads), 0, s, (const bf16_t*)dy, lddy, off_all, off_mask, pm,
               (bf16_t*)dx, (const bf16_t*)ref, H, W, C, OH, OW);
    return l2s_check_launch();
  }
  L2S_LAUNCH(adaptive_pool_bwd_kernel, dim3(cdiv(C, 256), H * W), dim3(256), 0, s, dy, lddy, off_all, off_mask, pm, dx, ref, H, W, C, OH, OW, dtype);
  return l2s_check_launch();
}
extern "C" int l2s_mask_downsample(const uint8_t* mask, float* out, int H, int W, int h, int w, hipStream_t s) {
  L2S_LAUNCH(mask_downsample_kernel, dim3(cdiv(h * w, 4)), dim3(256), 0, s, mask, out, H, W, h, w);
  return l2s_check_launch();
}
extern "C" int l2s_counter_inc(uint64_t* counter_dev, hipStream_t s) {
  L2S_LAUNCH(counter_inc_kernel, dim3(1), dim3(1), 0, s, counter_dev);
  return l2s_check_launch();
}
extern "C" int l2s_stamp(uint64_t* slot_dev, hipStream_t s) {
  L2S_LAUNCH(stamp_kernel, dim3(1), dim3(1), 0, s, slot_dev);
  return l2s_check_launch();
}
extern "C" int l2s_dropout_mask(float* mask, long n, float p, const uint64_t* seed_dev, uint64_t salt, hipStream_t s) {
  L2S_LAUNCH(dropout_kernel, dim3(grid_for(n)), dim3(256), 0, s, mask, n, p, seed_dev, salt);
  return l2s_check_launch();
}
extern "C" int l2s_random_keys(uint32_t* keys, long n, const uint64_t* seed_dev, uint64_t salt, hipStream_t s) {
  L2S_LAUNCH(keys_kernel, dim3(grid_for(n)), dim3(256), 0, s, keys, n, seed_dev, salt);
  return l2s_check_launch();
}
// persistent workgroups of the update: one per CU.  The update is HBM-bound either way (1.2 GB in ~0.3 ms); what the count decides is how
// deep the memory queues stand for everybody else: with 512 workgroups (25 MB of requests in flight) the next step's stem ended 475 us into the
// step and layer1 at 697 us; with 256 at 287 / 567 us while the update itself got no slower (409 against 430 us).  64 / 96 / 128 / 192 / 256 / 320 /
// 384 / 512 -> -- / 185.2 / 189.4 / 193.6 / 194.5 / 194.4 / 193.4 / 191.6 img/s, same box (profiles/r04_sgd_blocks.txt)
extern "C" int l2s_sgd_chunk(void) { return SGD_CHUNK; }
extern "C" int l2s_sgd_momentum(float* param, float* grad, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                                float lr, float momentum, float wd, float grad_scale, void* shadow, int shadow_dtype, int clear_grad, hipStream_t s) {
  if (nseg <= 0) return L2S_OK;
  L2S_LAUNCH(sgd_kernel, dim3(l2s_knobs::sgd_blocks), dim3(256), 0, s, param, grad, mom, segs, nseg, rowscale, lr, momentum, wd, grad_scale, shadow, shadow_dtype,
             clear_grad, 0L, (long)1 << 62, 0, -1, (const bf16_t*)nullptr, 0L);
  return l2s_check_launch();
}
extern "C" int l2s_sgd_momentum_range(float* param, float* grad, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                                      float lr, float momentum, float wd, float grad_scale, void* shadow, int shadow_dtype, int flags,
                                      long lo, long hi, int chunk_lo, int chunk_hi, hipStream_t s) {
  if (nseg <= 0 || hi <= lo) return L2S_OK;
  if (chunk_hi >= 0 && chunk_hi <= chunk_lo) return L2S_OK;
  int blocks = l2s_knobs::sgd_blocks;
  if (chunk_hi >= 0 && chunk_hi - chunk_lo < blocks) blocks = chunk_hi - chunk_lo;
  L2S_LAUNCH(sgd_kernel, dim3(blocks), dim3(256), 0, s, param, grad, mom, segs, nseg, rowscale, lr, momentum, wd, grad_scale, shadow, shadow_dtype, flags,
             lo, hi, chunk_lo, chunk_hi, (const bf16_t*)nullptr, 0L);
  return l2s_check_launch();
}
extern "C" int l2s_sgd_momentum_range_g16(float* param, const void* grad_bf16, long grad_lo, float* mom, const l2s_sgd_seg* segs, int nseg, const float* rowscale,
                                          float lr, float momentum, float wd, float grad_scale, void* shadow, int shadow_dtype,
                                          long lo, long hi, int chunk_lo, int chunk_hi, hipStream_t s) {
  if (nseg <= 0 || hi <= lo) return L2S_OK;
  if (chunk_hi >= 0 && chunk_hi <= chunk_lo) return L2S_OK;
  if (!grad_bf16 || lo < grad_lo || ((grad_lo & 3) != 0) || ((uintptr_t)grad_bf16 & 7)) return L2S_EINVAL;
  int blocks = l2s_knobs::sgd_blocks;
  if (chunk_hi >= 0 && chunk_hi - chunk_lo < blocks) blocks = chunk_hi - chunk_lo;
  L2S_LAUNCH(sgd_kernel, dim3(blocks), dim3(256), 0, s, param, (float*)nullptr, mom, segs, nseg, rowscale, lr, momentum, wd, grad_scale, shadow, shadow_dtype, 0,
             lo, hi, chunk_lo, chunk_hi, (const bf16_t*)grad_bf16, grad_lo);
  return l2s_check_launch();
}

#ifdef L2S_TOOLS
// ---- the tools build only (csrc/knobs.h): the tunables as variables, one setter.  Not compiled into liblang2seg_hip.so. ----
#include <string.h>
namespace l2s_knobs {
int pdma_wgs = 0, dma256_auto = 1, wgrad_grid_cap = 0, wgrad_row3_dma = 1, wgrad_row3_dma_wgs = 160, wgrad_row3_wide = 1, wgrad_row3_min_m = 8192,
    wgrad_1x1_dma = 0, row3_form = 0, row3_plan_mode = 0, sgd_blocks = 256;
}
// clock probe (tools/clock_probe.py): one wave runs a fixed dependent chain of 2048 integer multiply-adds; out[0] = its duration in ticks of the
// constant 100 MHz clock, out[1] = in s_memtime counts.  The chain's length in core cycles is fixed, so out[0] is inversely proportional
// to the core clock at that point of the step.
__global__ void clock_probe_kernel(uint64_t* out) {
  const uint64_t w0 = wall_clock64(), c0 = __builtin_amdgcn_s_memtime();
  unsigned int x = threadIdx.x + 1;
#pragma unroll 16
  for (int i = 0; i < 2048; ++i) { x = x * 1664525u + 1013904223u; asm volatile("" : "+v"(x)); }
  const uint64_t w1 = wall_clock64(), c1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) { out[0] = w1 - w0; out[1] = c1 - c0; out[2] = x; }
}
extern "C" int l2s_tools_clock_probe(uint64_t* out3, hipStream_t s) {
  L2S_LAUNCH(clock_probe_kernel, dim3(1), dim3(64), 0, s, out3);
  return l2s_check_launch();
}
extern "C" int l2s_tools_set(const char* name, int value) {
  using namespace l2s_knobs;
  struct { const char* n; int* p; } tab[] = {{"pdma_wgs", &pdma_wgs}, {"dma256_auto", &dma256_auto}, {"wgrad_grid_cap", &wgrad_grid_cap},
    {"wgrad_row3_dma", &wgrad_row3_dma}, {"wgrad_row3_dma_wgs", &wgrad_row3_dma_wgs}, {"wgrad_row3_wide", &wgrad_row3_wide},
    {"wgrad_row3_min_m", &wgrad_row3_min_m}, {"wgrad_1x1_dma", &wgrad_1x1_dma}, {"row3_form", &row3_form}, {"row3_plan_mode", &row3_plan_mode},
    {"sgd_blocks", &sgd_blocks}};
  for (auto& e : tab) if (!strcmp(e.n, name)) { *e.p = value; return L2S_OK; }
  return L2S_EINVAL;
}
#endif
