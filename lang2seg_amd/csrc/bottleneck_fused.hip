// A frozen 64-plane ResNet bottleneck behind its first 1x1 convolution, as ONE launch (bf16 mode, forward only):
//     b = relu(conv3x3(a; W2) + b2)            a = relu(bn(conv1(x))) of this block, [H*W][64]
//     y = relu(conv1x1(b; W3) + b3 + shortcut)  shortcut = x (Cx = 256) or conv1x1(x; Wd) + bd (Cx = 64: the first block of layer1)
//     a_next = relu(conv1x1(y; W1n) + b1n)      the NEXT block's first convolution, when the caller passes its weights
// (pyutils/mask-faster-rcnn/lib/nets/resnet_v1_cycle_res5_2.py:78-114 Bottleneck.forward, :291-299 the frozen layer1; BN folded into the
// bf16 weights and an f32 bias as everywhere in this library).
//
// Why: layer1 of the 600x1000 step is 10 launches over a 150x250 map whose three weight matrices are 140 KB per block and whose tensors are
// 19 MB: each launch round-trips 19-38 MB for a few GFLOP (227 us in all = 0.028 of peak, ~5x its HBM floor) at the head of every step,
// right where the previous step's update still owns the memory queues.  Nothing trains here, so nothing but y (and a_next) has to exist.
//
// Tile: 6 rows x 25 columns = 150 output pixels per workgroup (600x1000: 25 x 10 = 250 workgroups, one round on 256 CUs, no ragged
// edge), eight waves = two pixel halves (five 16-pixel fragments each) x four channel quarters.  WEIGHTS NEVER TOUCH LDS: a wave loads
// the MFMA fragments of its channel quarter straight from the [Cout][K] matrices into registers (72 + 32 (+ 32) + 32 VGPRs), with the
// operands swapped so that a lane owns four consecutive channels of a pixel.  LDS holds activations only:
//     AT  a with a one-pixel halo, 8 x 27 rows of 144 B (128 B of channels + 16 B of padding: tap (ky, kx) of a fragment is an
//         IMMEDIATE offset of its centre row, and 16 consecutive rows cover all 64 banks); later the staging area of a_next
//     BT  b, 160 rows of 144 B                         XT  x for the Cx = 64 shortcut, 160 rows of 144 B
//     YT  x (Cx = 256), then y in place, 160 rows of 528 B: the residual add reads and rewrites the lane's own 8 bytes; y leaves with
//         16-byte row-major stores and is the A operand of the next block's conv1 without another trip through HBM.
// Arithmetic: fp32 accumulation, bias / shortcut / ReLU in fp32, ONE rounding to bf16 per stored tensor - the rounding points of the
// unfused path, except that the first block's shortcut convolution is no longer rounded to bf16 before the add.
#include "common.h"
#include "../../include/lang2seg_hip.h"

namespace {

constexpr int TH = 6, TW = 25, TP = TH * TW;          // 150 output pixels per workgroup
constexpr int AW = TW + 2, AROWS = (TH + 2) * AW;     // halo tile: 8 x 27 = 216 rows
constexpr int RS = 144;                               // LDS row stride of the 64-channel tiles (bytes)
constexpr int YS = 528;                               // LDS row stride of the 256-channel tile
constexpr int MP = 160;                               // pixels padded to ten 16-row fragments
constexpr int AT_OFF = 0, BT_OFF = AROWS * RS, XT_OFF = BT_OFF + MP * RS, YT_OFF = XT_OFF + MP * RS;   // 0, 31104, 54144, 77184
constexpr int LDS_BYTES = YT_OFF + MP * YS;           // 161 664 (<= 163 840)

typedef unsigned int u32x2b __attribute__((ext_vector_type(2)));

__device__ __forceinline__ f32x4 mfma_bf16(const uint4& w, const uint4& px, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, w), __builtin_bit_cast(bf16x8, px), c, 0, 0, 0);
}
__device__ __forceinline__ u32x2b pack4(float a, float b, float c, float d) {
  u32x2b r;
  r.x = (uint32_t)f2bf(a) | ((uint32_t)f2bf(b) << 16);
  r.y = (uint32_t)f2bf(c) | ((uint32_t)f2bf(d) << 16);
  return r;
}

template <bool DOWN>
__global__ __launch_bounds__(512) void bottleneck64_kernel(const l2s_bottleneck64_desc p, int tiles_x) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int mh = wave >> 2, nq = wave & 3;             // pixel half, channel quarter
  const int fr = lane & 15, fg = lane >> 4;
  const int H = p.H, W = p.W;
  const int by = blockIdx.x / tiles_x, bx = blockIdx.x - by * tiles_x;
  const int gy0 = by * TH, gx0 = bx * TW;
  const bf16_t* a = (const bf16_t*)p.a;
  const bf16_t* x = (const bf16_t*)p.x;
  bf16_t* y = (bf16_t*)p.y;

  // ---- stage a (+ halo) into AT; request x ----
  for (int idx = tid; idx < AROWS * 8; idx += 512) {
    const int r = idx >> 3, ch = idx & 7;
    const int ay = r / AW, ax = r - ay * AW;
    const int gy = gy0 + ay - 1, gx = gx0 + ax - 1;
    uint4 v = make_uint4(0, 0, 0, 0);
    if (gy >= 0 && gy < H && gx >= 0 && gx < W) v = *(const uint4*)(a + ((long)gy * W + gx) * 64 + ch * 8);
    *(uint4*)(smem + AT_OFF + r * RS + ch * 16) = v;
  }
  constexpr int XCH = DOWN ? 8 : 32;                   // 16-byte chunks per pixel of x
  constexpr int XN = (TP * XCH + 511) / 512;           // per thread: 3 (Cx = 64) or 10 (Cx = 256)
  uint4 xr[XN];
#pragma unroll
  for (int k = 0; k < XN; ++k) {
    const int idx = tid + 512 * k, pp = idx / XCH, ch = idx - pp * XCH;
    const int ty = pp / TW, tx = pp - ty * TW, gy = gy0 + ty, gx = gx0 + tx;
    xr[k] = make_uint4(0, 0, 0, 0);
    if (pp < TP && gy < H && gx < W) xr[k] = *(const uint4*)(x + ((long)gy * W + gx) * (XCH * 8) + ch * 8);
  }
  // ---- weight fragments of this wave's channel quarter, straight into registers ----
  uint4 w2[18];
  {
    const bf16_t* r2 = (const bf16_t*)p.w2 + (long)(16 * nq + fr) * 576 + 8 * fg;
#pragma unroll
    for (int s = 0; s < 18; ++s) w2[s] = *(const uint4*)(r2 + 32 * s);
  }
  // the five fragments of this wave: pixel of lane fr, its centre row in AT
  int prow[5], pix[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) {
    const int pp = 16 * (5 * mh + i) + fr;
    const int q = pp < TP ? pp : 0;
    const int ty = q / TW, tx = q - ty * TW;
    pix[i] = pp;
    prow[i] = (ty * AW + tx) * RS + fg * 16;           // tap (0, 0) of the 3x3 window = halo row (ty, tx)
  }
  __syncthreads();

  // ---- conv2: 3x3, 64 -> 16 channels of this wave, 18 k-steps (tap-major, two 32-channel halves per tap) ----
  f32x4 acc2[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc2[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int tap = 0; tap < 9; ++tap) {
    const int toff = ((tap / 3) * AW + (tap % 3)) * RS;
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
      for (int i = 0; i < 5; ++i) {
        const uint4 fa = *(const uint4*)(smem + AT_OFF + prow[i] + toff + hf * 64);
        acc2[i] = mfma_bf16(w2[2 * tap + hf], fa, acc2[i]);
      }
  }
  // the other stages' fragments (requested here: their latency hides behind the stores and the barrier)
  uint4 w3[4][2], wd[DOWN ? 4 : 1][2], w1n[8];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      w3[j][s] = *(const uint4*)((const bf16_t*)p.w3 + (long)(64 * nq + 16 * j + fr) * 64 + 32 * s + 8 * fg);
      if constexpr (DOWN) wd[j][s] = *(const uint4*)((const bf16_t*)p.wd + (long)(64 * nq + 16 * j + fr) * 64 + 32 * s + 8 * fg);
    }
  const bool next = p.w1n != nullptr;
  if (next) {
#pragma unroll
    for (int s = 0; s < 8; ++s) w1n[s] = *(const uint4*)((const bf16_t*)p.w1n + (long)(16 * nq + fr) * 256 + 32 * s + 8 * fg);
  }
  {
    const f32x4 bb = *(const f32x4*)(p.b2 + 16 * nq + 4 * fg);
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const u32x2b pk = pack4(fmaxf(acc2[i][0] + bb[0], 0.f), fmaxf(acc2[i][1] + bb[1], 0.f), fmaxf(acc2[i][2] + bb[2], 0.f), fmaxf(acc2[i][3] + bb[3], 0.f));
      *(u32x2b*)(smem + BT_OFF + pix[i] * RS + (16 * nq + 4 * fg) * 2) = pk;
    }
  }
#pragma unroll
  for (int k = 0; k < XN; ++k) {
    const int idx = tid + 512 * k, pp = idx / XCH, ch = idx - pp * XCH;
    if (pp < MP) *(uint4*)(smem + (DOWN ? XT_OFF + pp * RS : YT_OFF + pp * YS) + ch * 16) = xr[k];
  }
  __syncthreads();

  // ---- conv3: 1x1, 64 -> 64 channels of this wave (+ the shortcut convolution of the first block into the same accumulators) ----
  f32x4 acc3[5][4];
#pragma unroll
  for (int i = 0; i < 5; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc3[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 2; ++s)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint4 fa = *(const uint4*)(smem + BT_OFF + pix[i] * RS + s * 64 + fg * 16);
#pragma unroll
      for (int j = 0; j < 4; ++j) acc3[i][j] = mfma_bf16(w3[j][s], fa, acc3[i][j]);
      if constexpr (DOWN) {
        const uint4 fx = *(const uint4*)(smem + XT_OFF + pix[i] * RS + s * 64 + fg * 16);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc3[i][j] = mfma_bf16(wd[j][s], fx, acc3[i][j]);
      }
    }
  // ---- bias, shortcut, ReLU in fp32; y in place of x in YT ----
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int c = 64 * nq + 16 * j + 4 * fg;
    f32x4 bb = *(const f32x4*)(p.b3 + c);
    if constexpr (DOWN) { const f32x4 b2_ = *(const f32x4*)(p.bd + c); bb[0] += b2_[0]; bb[1] += b2_[1]; bb[2] += b2_[2]; bb[3] += b2_[3]; }
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      char* at = smem + YT_OFF + pix[i] * YS + c * 2;
      float v0 = acc3[i][j][0] + bb[0], v1 = acc3[i][j][1] + bb[1], v2 = acc3[i][j][2] + bb[2], v3 = acc3[i][j][3] + bb[3];
      if constexpr (!DOWN) {
        const u32x2b xv = *(const u32x2b*)at;
        v0 += __uint_as_float(xv.x << 16); v1 += __uint_as_float(xv.x & 0xFFFF0000u); v2 += __uint_as_float(xv.y << 16); v3 += __uint_as_float(xv.y & 0xFFFF0000u);
      }
      *(u32x2b*)at = pack4(fmaxf(v0, 0.f), fmaxf(v1, 0.f), fmaxf(v2, 0.f), fmaxf(v3, 0.f));
    }
  }
  __syncthreads();

  // ---- y: row-major 16-byte stores ----
#pragma unroll
  for (int k = 0; k < 10; ++k) {
    const int idx = tid + 512 * k, pp = idx >> 5, ch = idx & 31;
    const int ty = pp / TW, tx = pp - ty * TW, gy = gy0 + ty, gx = gx0 + tx;
    if (pp < TP && gy < H && gx < W) *(uint4*)(y + ((long)gy * W + gx) * 256 + ch * 8) = *(const uint4*)(smem + YT_OFF + pp * YS + ch * 16);
  }
  if (!next) return;
  // ---- the next block's conv1: 1x1, 256 -> 16 channels of this wave, A = y from YT ----
  f32x4 acc1[5];
#pragma unroll
  for (int i = 0; i < 5; ++i) acc1[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int s = 0; s < 8; ++s)
#pragma unroll
    for (int i = 0; i < 5; ++i) {
      const uint4 fa = *(const uint4*)(smem + YT_OFF + pix[i] * YS + s * 64 + fg * 16);
      acc1[i] = mfma_bf16(w1n[s], fa, acc1[i]);
    }
  {
    const f32x4 bb = *(const f32x4*)(p.b1n + 16 * nq + 4 * fg);
#pragma unroll
    for (int i = 0; i < 5; ++i)
      *(u32x2b*)(smem + AT_OFF + pix[i] * RS + (16 * nq + 4 * fg) * 2) =
          pack4(fmaxf(acc1[i][0] + bb[0], 0.f), fmaxf(acc1[i][1] + bb[1], 0.f), fmaxf(acc1[i][2] + bb[2], 0.f), fmaxf(acc1[i][3] + bb[3], 0.f));
  }
  __syncthreads();
  bf16_t* an = (bf16_t*)p.a_next;
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int idx = tid + 512 * k, pp = idx >> 3, ch = idx & 7;
    const int ty = pp / TW, tx = pp - ty * TW, gy = gy0 + ty, gx = gx0 + tx;
    if (pp < TP && gy < H && gx < W) *(uint4*)(an + ((long)gy * W + gx) * 64 + ch * 8) = *(const uint4*)(smem + AT_OFF + pp * RS + ch * 16);
  }
}

}  // namespace

extern "C" int l2s_bottleneck64_fwd(const l2s_bottleneck64_desc* d, hipStream_t s) {
  if (!d || !d->a || !d->x || !d->y || !d->w2 || !d->w3 || !d->b2 || !d->b3 || d->H < 1 || d->W < 1) return L2S_EINVAL;
  const bool down = d->Cx == 64;
  if (!(down || d->Cx == 256) || (down && (!d->wd || !d->bd)) || (d->w1n && (!d->b1n || !d->a_next))) return L2S_EINVAL;
  if ((long)d->H * d->W * 512 >= (1L << 31)) return L2S_EINVAL;
  const int tx = cdiv(d->W, TW), ty = cdiv(d->H, TH);
  const l2s_bottleneck64_desc v = *d;
  static bool attr_done = false;
  if (!attr_done) {
    (void)hipFuncSetAttribute((const void*)bottleneck64_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    (void)hipFuncSetAttribute((const void*)bottleneck64_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
    attr_done = true;
  }
  if (down) L2S_LAUNCH(bottleneck64_kernel<true>, dim3(tx * ty), dim3(512), (size_t)LDS_BYTES, s, v, tx);
  else L2S_LAUNCH(bottleneck64_kernel<false>, dim3(tx * ty), dim3(512), (size_t)LDS_BYTES, s, v, tx);
  return l2s_check_launch();
}
